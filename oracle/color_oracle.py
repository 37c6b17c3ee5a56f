"""TEST INFRASTRUCTURE - CPU restatement (numpy, C float / double semantics emulated dtype by dtype) of the colour
augmentations of the reference's train transform, used only by tests/ as the checker of clipa_amd/csrc/augment.hip.

clipa_torch/open_clip/transform.py:61-84,160-168 applies, on PIL images, torchvision ColorJitter(brightness, contrast,
saturation, hue) with probability color_jitter_prob and Grayscale(3) with probability gray_scale_prob (the GPU recipes use
0.32 / 0.32 / 0.32 / 0.08, p = 0.8 and p = 0.2: scripts/exp/gpu/vit_l16/i37_t8_pretrain.sh:13).  On PIL images those are
Pillow (third-party, not vendored by the reference; 12.2.0 here) code paths, restated below from the published sources:
  brightness / contrast / saturation = ImageEnhance.{Brightness,Contrast,Color}(img).enhance(f) = Image.blend(degenerate, img, f)
      with degenerate = black / solid gray int(mean(L) + 0.5) / L replicated, L = rgb2l (see resize_oracle.grayscale3);
      libImaging Blend.c: out = (UINT8)(in1 + alpha * (in2 - in1)) in C float, clipped to [0, 255] when alpha is outside [0, 1];
  hue = torchvision F.adjust_hue: convert('HSV'), h += uint8(hue_factor * 255) (wrapping), convert('RGB')
      (libImaging Convert.c rgb2hsv_row / hsv2rgb, the colorsys formulas in mixed float / double arithmetic).
torchvision applies the four in a random order (torch.randperm(4)) with factors ~ U(max(0, 1 - x), 1 + x), hue ~ U(-h, h).
PINNED: tests/test_augment_cpu.py checks every function bit for bit against Pillow itself and against
tests/golden/color_jitter_pil.npz (oracle/make_augment_golden.py)."""
import numpy as np

from .resize_oracle import grayscale3  # noqa: F401  (re-exported: the full transform's last stage)

f32, f64 = np.float32, np.float64
BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3


def rgb2l(img):
    v = img.astype(np.int64)
    return ((v[..., 0] * 19595 + v[..., 1] * 38470 + v[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(deg, img, alpha):
    """Blend.c ImagingBlend(im1 = degenerate, im2 = image, alpha) on uint8 arrays; alpha is a C float."""
    a = f32(alpha)
    if a == f32(0.0):
        return deg.copy()
    if a == f32(1.0):
        return img.copy()
    i1, i2 = deg.astype(np.int32), img.astype(np.int32)
    temp = (i1.astype(f32) + a * (i2 - i1).astype(f32)).astype(f32)
    if f32(0.0) <= a <= f32(1.0):
        return temp.astype(np.int32).astype(np.uint8)                      # (UINT8) cast: truncation
    return np.where(temp <= 0, 0, np.where(temp >= 255, 255, temp.astype(np.int32))).astype(np.uint8)


def adjust_brightness(img, f):
    return blend(np.zeros_like(img), img, f)


def adjust_contrast(img, f):
    l = rgb2l(img)
    mean = int(float(int(l.astype(np.int64).sum())) / l.size + 0.5)         # ImageStat.Stat(L).mean[0], rounded as ImageEnhance does
    return blend(np.full_like(img, mean), img, f)


def adjust_saturation(img, f):
    return blend(np.repeat(rgb2l(img)[..., None], 3, -1), img, f)


def rgb2hsv(img):
    r, g, b = (img[..., i].astype(np.int32) for i in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = (cr / maxc.astype(f32)).astype(f32)
        rc, gc, bc = (((maxc - c).astype(f32) / cr).astype(f32) for c in (r, g, b))
        h = np.where(r == maxc, (bc - gc).astype(f32),
                     np.where(g == maxc, (f64(2.0) + rc.astype(f64) - bc.astype(f64)).astype(f32),
                              (f64(4.0) + gc.astype(f64) - rc.astype(f64)).astype(f32)))
        h = np.fmod(h.astype(f64) / 6.0 + 1.0, 1.0).astype(f32)
        uh = np.clip(np.nan_to_num(h.astype(f64) * 255.0).astype(np.int64), 0, 255)
        us = np.clip(np.nan_to_num(s.astype(f64) * 255.0).astype(np.int64), 0, 255)
    gray = minc == maxc
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def _c_round(x):
    return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5))             # C round(): halves away from zero


def hsv2rgb(hsv):
    h, s, v = (hsv[..., i] for i in range(3))
    hf = h.astype(f32).astype(f64) * 6.0 / 255.0
    i = np.floor(hf).astype(np.int64)
    f = (hf - i.astype(f32).astype(f64)).astype(f32)
    fs = (s.astype(f32).astype(f64) / 255.0).astype(f32)
    vf = v.astype(f32).astype(f64)
    p = _c_round(vf * (1.0 - fs.astype(f64)))
    q = _c_round(vf * (1.0 - fs.astype(f64) * f.astype(f64)))
    t = _c_round(vf * (1.0 - fs.astype(f64) * (1.0 - f.astype(f64))))
    p, q, t = (np.clip(x, 0, 255).astype(np.uint8) for x in (p, q, t))
    k = i % 6
    r = np.choose(k, [v, q, p, p, t, v])
    g = np.choose(k, [t, v, v, q, p, p])
    b = np.choose(k, [p, p, t, v, v, q])
    gray = s == 0
    return np.stack([np.where(gray, v, r), np.where(gray, v, g), np.where(gray, v, b)], -1).astype(np.uint8)


def hue_shift(hue_factor):
    """torchvision adjust_hue's np.uint8(hue_factor * 255): truncation toward zero, modulo 256."""
    return int(hue_factor * 255) & 0xFF


def adjust_hue(img, hue_factor):
    hsv = rgb2hsv(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift(hue_factor)).astype(np.uint8)
    return hsv2rgb(hsv)


_OPS = {BRIGHTNESS: adjust_brightness, CONTRAST: adjust_contrast, SATURATION: adjust_saturation, HUE: adjust_hue}


def color_jitter(img, order, factors):
    """torchvision ColorJitter.forward with the sampled permutation `order` (4 op ids) and `factors` indexed by op id
    (brightness, contrast, saturation, hue)."""
    for op in order:
        img = _OPS[int(op)](img, float(factors[int(op)]))
    return img
