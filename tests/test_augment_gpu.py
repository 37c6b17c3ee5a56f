"""GPU parity of the device-side input pipeline (clipa_resized_crop_u8 through the C ABI, clipa_amd.data.DeviceAugment /
DevicePrefetcher) - integer / byte work, so the bar is BIT-EXACT: against the Pillow-generated fixture, against the numpy
restatement of Pillow's resampler (oracle/resize_oracle.py) on random crops, and against the live Pillow when importable."""
import os

import numpy as np
import pytest
import torch

from clipa_amd import data as D
from oracle import color_oracle as C
from oracle import resize_oracle as R

from .conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ops():
    from clipa_amd import ops as _ops
    return _ops


def test_resized_crop_matches_pillow_fixture():
    z = np.load(os.path.join(GOLDEN, "resized_crop_pil.npz"))
    src = torch.from_numpy(z["images"]).to(DEV)
    boxes = torch.from_numpy(z["boxes"]).to(DEV)
    for S in (32, 56):
        out = ops().resized_crop_u8(src, boxes, S).cpu().numpy()
        assert np.array_equal(out, z[f"resized_{S}"]), f"size {S}: {(out != z[f'resized_{S}']).mean():.5f} of the bytes differ"
    flags = torch.tensor([1, 0, 1, 0, 0, 1], dtype=torch.uint8, device=DEV)
    out = ops().resized_crop_u8(src, boxes, 56, flags).cpu().numpy()
    for i in range(6):
        ref = z["resized_56"][i]
        assert np.array_equal(out[i], R.grayscale3(ref) if int(flags[i]) else ref), i
    ops().check_token_ids(wait=True)


@pytest.mark.parametrize("Hs,Ws,S", [(160, 160, 84), (96, 200, 112), (256, 256, 224), (300, 260, 64)])
def test_resized_crop_random_boxes_vs_oracle(Hs, Ws, S):
    B = 12
    g = torch.Generator().manual_seed(Hs + S)
    src = torch.randint(0, 256, (B, Hs, Ws, 3), generator=g, dtype=torch.uint8)
    boxes = D.sample_crop_boxes(B, Hs, Ws, (0.08, 1.0), (3 / 4, 4 / 3), g)
    boxes[0] = torch.tensor([0, 0, Hs, Ws], dtype=torch.int32)                # the whole image
    boxes[1] = torch.tensor([Hs - 9, Ws - 11, 9, 11], dtype=torch.int32)      # a tiny corner crop: strong up-scaling
    out = ops().resized_crop_u8(src.to(DEV), boxes.to(DEV), S).cpu().numpy()
    try:
        from PIL import Image
    except ImportError:
        Image = None
    for i in range(B):
        t, l, h, w = (int(v) for v in boxes[i])
        ref = R.resized_crop(src[i].numpy(), t, l, h, w, S)
        assert np.array_equal(out[i], ref), (i, t, l, h, w, float((out[i] != ref).mean()))
        if Image is not None and i < 4:
            pil = np.asarray(Image.fromarray(src[i].numpy()).crop((l, t, l + w, t + h)).resize((S, S), Image.BICUBIC))
            assert np.array_equal(out[i], pil), ("pillow", i)
    ops().check_token_ids(wait=True)


def test_resized_crop_identity_and_batch_scale():
    """Size-independent property at the bench's shape (4096 staged 256 x 256 images -> 224): a full-image box at the source
    size returns the image unchanged; sampled outputs of the big batch equal the oracle."""
    B, Hs, S = 4096, 256, 224
    g = torch.Generator().manual_seed(7)
    base = torch.randint(0, 256, (64, Hs, Hs, 3), generator=g, dtype=torch.uint8).to(DEV)
    src = base.repeat(B // 64, 1, 1, 1).contiguous()
    full = torch.tensor([[0, 0, Hs, Hs]], dtype=torch.int32).repeat(B, 1).to(DEV)
    same = ops().resized_crop_u8(src, full, Hs)
    assert torch.equal(same, src)
    boxes = D.sample_crop_boxes(B, Hs, Hs, (0.4, 1.0), (3 / 4, 4 / 3), g)
    out = ops().resized_crop_u8(src, boxes.to(DEV), S)
    for i in (0, 1, 777, 2048, 4095):
        t, l, h, w = (int(v) for v in boxes[i])
        assert np.array_equal(out[i].cpu().numpy(), R.resized_crop(src[i].cpu().numpy(), t, l, h, w, S)), i
    ops().check_token_ids(wait=True)


def test_resized_crop_rejects_bad_boxes():
    src = torch.randint(0, 256, (3, 64, 64, 3), dtype=torch.uint8).to(DEV)
    boxes = torch.tensor([[0, 0, 64, 64], [10, 10, 60, 20], [0, 0, 32, 32]], dtype=torch.int32).to(DEV)   # sample 1 leaves the image
    out = ops().resized_crop_u8(src, boxes, 16)
    torch.cuda.synchronize()
    assert int(out[1].max()) == 0 and int(out[0].max()) > 0 and int(out[2].max()) > 0
    with pytest.raises(RuntimeError, match="rejected"):
        ops().check_token_ids(wait=True)


def test_device_augment_and_prefetcher():
    S, Hs = 56, 72
    g = torch.Generator().manual_seed(1)
    batches = [(torch.randint(0, 256, (8, Hs, Hs, 3), generator=g, dtype=torch.uint8), torch.full((8, 5), i, dtype=torch.int64))
               for i in range(4)]
    aug = D.DeviceAugment(S, scale=(0.4, 1.0), gray_scale_prob=0.5, seed=3)
    ref_gen = torch.Generator().manual_seed(3)
    got = list(D.DevicePrefetcher(iter(batches), DEV, transform=aug, depth=2))
    assert len(got) == 4
    for i, (img, txt) in enumerate(got):
        assert img.shape == (8, 3, S, S) and img.dtype == torch.uint8 and img.is_contiguous(memory_format=torch.channels_last)
        assert int(txt[0, 0]) == i and txt.is_cuda
        boxes = D.sample_crop_boxes(8, Hs, Hs, (0.4, 1.0), (3 / 4, 4 / 3), ref_gen)      # the augmenter's own random stream
        gray = (torch.rand(8, generator=ref_gen) < 0.5)
        for j in range(8):
            t, l, h, w = (int(v) for v in boxes[j])
            ref = R.resized_crop(batches[i][0][j].numpy(), t, l, h, w, S)
            if bool(gray[j]):
                ref = R.grayscale3(ref)
            assert np.array_equal(img[j].permute(1, 2, 0).cpu().numpy(), ref), (i, j)
    ops().check_token_ids(wait=True)


def test_color_jitter_matches_pillow_fixture():
    z = np.load(os.path.join(GOLDEN, "color_jitter_pil.npz"))
    img = torch.from_numpy(z["images"]).to(DEV).clone()
    out = ops().color_jitter_u8_(img, None, torch.from_numpy(z["orders"]).to(DEV), torch.from_numpy(z["factors"]).to(DEV))
    got = out.cpu().numpy()
    for i in range(len(got)):
        assert np.array_equal(got[i], z["jittered"][i]), (i, z["orders"][i], float((got[i] != z["jittered"][i]).mean()))


def test_color_jitter_random_batch_vs_oracle():
    B, S = 48, 56
    g = torch.Generator().manual_seed(11)
    img = torch.randint(0, 256, (B, S, S, 3), generator=g, dtype=torch.uint8)
    img[1, ..., 1] = img[1, ..., 0]
    img[2] = 200
    img[3] = (img[3] // 32) * 32
    apply, order, factors = D.sample_color_jitter(B, 0.32, 0.32, 0.32, 0.08, 0.8, g)
    factors[5] = torch.tensor([1.0, 1.0, 1.0, 0.0])
    factors[6] = torch.tensor([0.0, 2.5, 0.0, -0.5])                  # extrapolating blends, the largest hue shift
    gray = (torch.rand(B, generator=g) < 0.3).to(torch.uint8)
    out = ops().color_jitter_u8_(img.to(DEV).clone(), apply.to(DEV), order.to(DEV), factors.to(DEV), gray.to(DEV)).cpu().numpy()
    for i in range(B):
        ref = img[i].numpy()
        if int(apply[i]):
            ref = C.color_jitter(ref, order[i].numpy(), factors[i].numpy())
        if int(gray[i]):
            ref = R.grayscale3(ref)
        assert np.array_equal(out[i], ref), (i, int(apply[i]), order[i].tolist(), factors[i].tolist(), float((out[i] != ref).mean()))
    only_gray = ops().color_jitter_u8_(img.to(DEV).clone(), gray_flags=gray.to(DEV)).cpu().numpy()
    for i in range(B):
        assert np.array_equal(only_gray[i], R.grayscale3(img[i].numpy()) if int(gray[i]) else img[i].numpy())


def test_device_augment_full_recipe():
    """The reference GPU recipe (scripts/exp/gpu/vit_l16/i37_t8_pretrain.sh:13): scale (0.4, 1), color_jitter (0.32, 0.32, 0.32,
    0.08) p = 0.8, gray_scale p = 0.2 - resize, jitter and grayscale composed on the device equal the oracle's composition."""
    S, Hs = 48, 64
    g = torch.Generator().manual_seed(2)
    staged = torch.randint(0, 256, (16, Hs, Hs, 3), generator=g, dtype=torch.uint8)
    kw = dict(scale=(0.4, 1.0), color_jitter=(0.32, 0.32, 0.32, 0.08), color_jitter_prob=0.8, gray_scale_prob=0.2)
    out = D.DeviceAugment(S, seed=9, **kw)(staged.to(DEV))
    boxes, (apply, order, factors), gray = D.DeviceAugment(S, seed=9, **kw).sample(16, Hs, Hs)     # the same random stream
    assert out.shape == (16, 3, S, S) and out.is_contiguous(memory_format=torch.channels_last)
    for i in range(16):
        t, l, h, w = (int(v) for v in boxes[i])
        ref = R.resized_crop(staged[i].numpy(), t, l, h, w, S)
        if int(apply[i]):
            ref = C.color_jitter(ref, order[i].numpy(), factors[i].numpy())
        if int(gray[i]):
            ref = R.grayscale3(ref)
        assert np.array_equal(out[i].permute(1, 2, 0).cpu().numpy(), ref), i
    ops().check_token_ids(wait=True)


def test_model_consumes_augmented_batch():
    """The augmenter's channels_last uint8 output is a valid model input (normalised inside the patch gather)."""
    import clipa_amd
    from .conftest import load_golden
    gd = load_golden("cls_erf")
    m = clipa_amd.CLIP(**gd.cfg, output_dict=True)
    m.load_state_dict(gd.sd)
    m.to(DEV)
    S = gd.cfg["vision_cfg"]["image_size"]
    staged = gd.images_u8.permute(0, 2, 3, 1).contiguous().to(DEV)          # NHWC at the model's own size
    full = torch.tensor([[0, 0, S, S]], dtype=torch.int32).repeat(staged.shape[0], 1).to(DEV)
    x = ops().resized_crop_u8(staged, full, S).permute(0, 3, 1, 2)          # identity crop -> same pixels, NHWC memory
    with torch.no_grad():
        a = m.encode_image(x, normalize=True)
        b = m.encode_image(gd.images_u8.to(DEV), normalize=True)
    assert torch.equal(a, b)
