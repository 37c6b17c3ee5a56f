"""Synthetic LAION-shaped batches for benchmarks and smoke runs (the reference's own SyntheticDataset,
clipa_torch/training/data.py:469-486, yields an all-black image - useless for numerics and power).

images: uint8 [B, 3, S, S] uniform 0..255 - the `--to-float-on-device` wire format (open_clip/transform.py:171-174,
training/train.py:191-197) - returned in channels_last memory (physical NHWC: what a decoder produces and what the
patch gather reads in contiguous 3*P-byte runs).  texts: int64 [B, ctx] rows [SOT, k random ids, EOT, 0 ...] with
caption length k+2 ~ clip(N(20, 8), 3, ctx); EOT is the row maximum because pooling is text.argmax(-1) (model.py:254)."""
import torch


def synthetic_batch(batch, image_size, ctx, vocab, seed, device="cpu", channels_last=True):
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    base = min(batch, 256)                                     # host-side generation stays small; tiled on the device
    img = torch.randint(0, 256, (base, 3, image_size, image_size), generator=g, dtype=torch.uint8)
    sot, eot = vocab - 2, vocab - 1
    n = torch.clamp(torch.round(torch.randn(base, generator=g) * 8 + min(20.0, ctx * 0.6)), 3, ctx).long()
    txt = torch.randint(1, vocab - 2, (base, ctx), generator=g, dtype=torch.int64)
    pos = torch.arange(ctx).unsqueeze(0)
    txt = torch.where(pos < (n - 1).unsqueeze(1), txt, torch.zeros_like(txt))
    txt[:, 0] = sot
    txt[torch.arange(base), n - 1] = eot
    img, txt = img.to(device), txt.to(device)
    reps = (batch + base - 1) // base
    img = img.repeat(reps, 1, 1, 1)[:batch]
    txt = txt.repeat(reps, 1)[:batch].contiguous()
    # de-duplicate the tiled rows so no two pairs of the batch are identical
    img = img + (torch.arange(batch, device=img.device) % 251).to(torch.uint8).view(batch, 1, 1, 1)
    txt[:, 1] = 1 + (torch.arange(batch, device=txt.device) % (vocab - 3))
    img = img.contiguous(memory_format=torch.channels_last) if channels_last else img.contiguous()
    return img, txt
