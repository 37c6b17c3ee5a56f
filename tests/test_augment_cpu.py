"""CPU tests of the device-side input pipeline's checker and host logic: the numpy restatement of
Pillow's bicubic resize / rgb2l (oracle/resize_oracle.py) against the Pillow-generated fixture and against the live Pillow,
the crop-box sampler against torchvision's published get_params, and the prefetcher's ordering / buffering on the CPU."""
import os

import numpy as np
import pytest
import torch

from clipa_amd import data as D
from oracle import color_oracle as C
from oracle import resize_oracle as R

from .conftest import GOLDEN


def test_resize_oracle_matches_pillow_fixture():
    z = np.load(os.path.join(GOLDEN, "resized_crop_pil.npz"))
    for S in (32, 56):
        for i, (t, l, h, w) in enumerate(z["boxes"]):
            got = R.resized_crop(z["images"][i], int(t), int(l), int(h), int(w), S)
            assert np.array_equal(got, z[f"resized_{S}"][i]), (S, i)
    for i in range(len(z["images"])):
        assert np.array_equal(R.grayscale3(z["images"][i]), z["gray"][i])


def test_resize_oracle_matches_live_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    for trial in range(12):
        H, W = int(rng.integers(40, 200)), int(rng.integers(40, 200))
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        h, w = int(rng.integers(8, H + 1)), int(rng.integers(8, W + 1))
        t, l = int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1))
        S = int(rng.choice([24, 84, 112, w]))
        ref = np.asarray(Image.fromarray(img).crop((l, t, l + w, t + h)).resize((S, S), Image.BICUBIC))
        assert np.array_equal(R.resized_crop(img, t, l, h, w, S), ref), (trial, H, W, t, l, h, w, S)


def test_color_oracle_matches_pillow_fixture():
    """All 24 orders of brightness / contrast / saturation / hue, produced by Pillow's ImageEnhance and HSV conversions."""
    z = np.load(os.path.join(GOLDEN, "color_jitter_pil.npz"))
    for i in range(len(z["images"])):
        got = C.color_jitter(z["images"][i], z["orders"][i], z["factors"][i])
        assert np.array_equal(got, z["jittered"][i]), (i, z["orders"][i])


def test_color_oracle_matches_live_pillow():
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageEnhance
    rng = np.random.default_rng(5)
    for trial in range(40):
        img = rng.integers(0, 256, (int(rng.integers(6, 30)), int(rng.integers(6, 30)), 3), dtype=np.uint8)
        if trial % 4 == 0:
            img[..., 2] = img[..., 1]
        pil = Image.fromarray(img)
        f = float(rng.uniform(0.5, 1.5))
        assert np.array_equal(C.adjust_brightness(img, f), np.asarray(ImageEnhance.Brightness(pil).enhance(f)))
        assert np.array_equal(C.adjust_contrast(img, f), np.asarray(ImageEnhance.Contrast(pil).enhance(f)))
        assert np.array_equal(C.adjust_saturation(img, f), np.asarray(ImageEnhance.Color(pil).enhance(f)))
        assert np.array_equal(C.rgb2hsv(img), np.asarray(pil.convert("HSV")))
        assert np.array_equal(C.hsv2rgb(img), np.asarray(Image.fromarray(img, "HSV").convert("RGB")))


def test_color_jitter_sampler():
    g = torch.Generator().manual_seed(0)
    apply, order, factors = D.sample_color_jitter(5000, 0.32, 0.32, 0.32, 0.08, 0.8, g)
    assert apply.dtype == torch.uint8 and 0.77 < apply.float().mean() < 0.83
    assert (order.sort(1).values == torch.arange(4, dtype=torch.int32)).all()          # every row a permutation
    first = torch.bincount(order[:, 0].long(), minlength=4).float() / 5000
    assert (first - 0.25).abs().max() < 0.03                                           # uniformly random order
    assert factors[:, :3].min() >= 0.68 - 1e-6 and factors[:, :3].max() <= 1.32 + 1e-6
    assert factors[:, 3].min() >= -0.08 - 1e-6 and factors[:, 3].max() <= 0.08 + 1e-6
    assert abs(float(factors[:, 0].mean()) - 1.0) < 0.01 and abs(float(factors[:, 3].mean())) < 0.003


def test_crop_box_sampler():
    g = torch.Generator().manual_seed(0)
    scale, ratio = (0.4, 1.0), (3 / 4, 4 / 3)
    b = D.sample_crop_boxes(20000, 256, 256, scale, ratio, g).long()
    t, l, h, w = b.T
    assert (t >= 0).all() and (l >= 0).all() and (t + h <= 256).all() and (l + w <= 256).all() and (h > 0).all() and (w > 0).all()
    frac = (h * w).double() / 65536
    assert frac.min() > 0.39 and frac.max() <= 1.0
    asp = w.double() / h.double()
    assert asp.min() > 0.74 and asp.max() < 1.35                       # the ratio bounds up to integer rounding
    seq = torch.tensor([D.random_resized_crop_params(256, 256, scale, ratio, g) for _ in range(4000)]).double()
    # the vectorised sampler draws from the same distribution as the sequential restatement of get_params
    assert abs(seq[:, 2].mean() - h.double().mean()) < 2.0 and abs(seq[:, 3].mean() - w.double().mean()) < 2.0
    assert abs(seq[:, 0].mean() - t.double().mean()) < 1.5 and abs((seq[:, 2] * seq[:, 3]).mean() / 65536 - frac.mean()) < 0.01
    # an image far outside the ratio bounds: every attempt fails -> the central fallback crop
    fb = D.sample_crop_boxes(50, 64, 512, (0.9, 1.0), ratio, g)
    assert (fb == torch.tensor(D.random_resized_crop_params(64, 512, (2.0, 2.0), ratio), dtype=torch.int32)).all()
    assert tuple(fb[0].tolist()) == (0, 213, 64, 85)


def test_prefetcher_order_and_depth_on_cpu():
    batches = [(torch.full((2, 4, 4, 3), i, dtype=torch.uint8), torch.full((2, 5), i, dtype=torch.int64)) for i in range(5)]
    seen = []

    def loader():
        for b in batches:
            seen.append(int(b[1][0, 0]))
            yield b

    out = []
    for img, txt in D.DevicePrefetcher(loader(), "cpu", transform=lambda x: x + 1, depth=2):
        out.append((int(img[0, 0, 0, 0]), int(txt[0, 0]), len(seen)))
    assert [o[:2] for o in out] == [(i + 1, i) for i in range(5)]
    assert out[0][2] == 2 and out[1][2] == 3                           # runs `depth` batches ahead of the consumer
