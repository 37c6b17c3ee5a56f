"""Generate tests/golden/resized_crop_pil.npz from Pillow itself - the library torchvision's RandomResizedCrop / Grayscale
call on the PIL images of the reference's train transform (clipa_torch/open_clip/transform.py:152-168).  Pillow is not
vendored under /root/reference (third-party, unpinned there); the version that produced the fixture is stored in it.
    python oracle/make_augment_golden.py
Small on purpose (a few 96 x 128 images, crops resized to 32 / 56): the fixture pins oracle/resize_oracle.py and, through
it, the HIP kernels; the tests additionally compare with the live Pillow whenever it is importable."""
import os

import numpy as np
import PIL
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rng = np.random.default_rng(20260926)
    H, W, N = 96, 128, 6
    imgs = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    imgs[0] = np.stack([yy * 255 // H, xx * 255 // W, (yy + xx) * 255 // (H + W)], -1).astype(np.uint8)   # smooth ramps
    imgs[1] = ((yy // 4 + xx // 4) % 2 * 255)[..., None].repeat(3, -1).astype(np.uint8)                      # checkerboard
    boxes = np.array([[0, 0, 96, 128], [10, 20, 40, 64], [30, 5, 66, 50], [0, 64, 96, 64], [7, 3, 32, 32], [50, 100, 20, 28]],
                     np.int32)                                                                              # top, left, h, w
    out = {}
    for S in (32, 56):
        res = np.zeros((N, S, S, 3), np.uint8)
        for i in range(N):
            t, l, h, w = (int(v) for v in boxes[i])
            res[i] = np.asarray(Image.fromarray(imgs[i]).crop((l, t, l + w, t + h)).resize((S, S), Image.BICUBIC))
        out[f"resized_{S}"] = res
    gray = np.stack([np.asarray(Image.fromarray(im).convert("L").convert("RGB")) for im in imgs])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "resized_crop_pil.npz"), images=imgs, boxes=boxes, gray=gray,
                        pillow_version=np.array(PIL.__version__), **out)
    print("wrote resized_crop_pil.npz with Pillow", PIL.__version__)
    color_jitter_fixture(rng)


def color_jitter_fixture(rng):
    """ImageEnhance / HSV paths of torchvision's ColorJitter on PIL images: every one of the 24 op orders once."""
    import itertools
    from PIL import ImageEnhance
    H = W = 40
    orders = np.array(list(itertools.permutations(range(4))), np.int32)          # 0 brightness 1 contrast 2 saturation 3 hue
    N = len(orders)
    imgs = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    imgs[1, ..., 1] = imgs[1, ..., 0]                                                # channel ties
    imgs[2] = imgs[2, 0, 0]                                                          # flat image
    imgs[3] = (imgs[3] // 64) * 64                                                   # few levels
    factors = np.stack([rng.uniform(0.68, 1.32, N), rng.uniform(0.68, 1.32, N), rng.uniform(0.68, 1.32, N),
                        rng.uniform(-0.08, 0.08, N)], 1).astype(np.float32)
    factors[4, :3] = 1.0
    factors[5] = (0.0, 0.0, 0.0, 0.0)
    out = np.zeros_like(imgs)
    for i in range(N):
        im = Image.fromarray(imgs[i])
        for op in orders[i]:
            f = float(factors[i, op])
            if op == 0:
                im = ImageEnhance.Brightness(im).enhance(f)
            elif op == 1:
                im = ImageEnhance.Contrast(im).enhance(f)
            elif op == 2:
                im = ImageEnhance.Color(im).enhance(f)
            else:                                                                    # torchvision F.adjust_hue
                h, s, v = im.convert("HSV").split()
                np_h = (np.array(h, dtype=np.uint8).astype(np.int32) + (int(f * 255) & 0xFF)).astype(np.uint8)
                im = Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")
        out[i] = np.asarray(im)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "color_jitter_pil.npz"), images=imgs, orders=orders, factors=factors,
                        jittered=out, pillow_version=np.array(PIL.__version__))
    print("wrote color_jitter_pil.npz")


if __name__ == "__main__":
    main()
