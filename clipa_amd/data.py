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


# ---------------------------------------------------------------------------------------------------------------------
# Device-side input pipeline.  The reference decodes, crops, resizes and colour-converts every image
# on host CPU workers (open_clip/transform.py:152-168 inside the loaders of training/data.py:332-436 /
# reader_tfds.py:322-350) and ships the finished uint8 tensor (`images.to(device, non_blocking=True)`,
# training/train.py:187-189); it already moved the float conversion to the device because the hosts could not keep up at
# ~9 k pairs/s.  Here the loader only has to deliver decoded uint8 NHWC images at a fixed staging resolution; the copy is
# double-buffered through pinned memory on its own HIP stream and RandomResizedCrop / Grayscale run as HIP kernels
# (clipa_amd/csrc/augment.hip, bit-exact with the Pillow code torchvision calls) while the previous step computes.
import collections
import math


def random_resized_crop_params(height, width, scale=(0.9, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), generator=None):
    """torchvision.transforms.RandomResizedCrop.get_params (the box sampler behind open_clip/transform.py:153-157),
    restated with an explicit torch.Generator: up to 10 attempts at (area fraction ~ U(scale), aspect ~ log-U(ratio)) that
    fit the image, else the largest central crop inside the ratio bounds.  -> (top, left, h, w)."""
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1], generator=generator).item()
        aspect = math.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1], generator=generator).item())
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = int(torch.randint(0, height - h + 1, (1,), generator=generator).item())
            j = int(torch.randint(0, width - w + 1, (1,), generator=generator).item())
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def sample_crop_boxes(batch, height, width, scale=(0.9, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), generator=None):
    """The same distribution for a whole batch in a handful of tensor ops (a Python loop over 4096 samples would cost more
    host time than the step it feeds): int32 [batch, 4] = (top, left, h, w) on the CPU."""
    area = float(height * width)
    u = torch.rand(batch, 10, generator=generator, dtype=torch.float64)
    v = torch.rand(batch, 10, generator=generator, dtype=torch.float64)
    target = area * (scale[0] + (scale[1] - scale[0]) * u)
    aspect = torch.exp(math.log(ratio[0]) + (math.log(ratio[1]) - math.log(ratio[0])) * v)
    w = torch.round(torch.sqrt(target * aspect)).long()
    h = torch.round(torch.sqrt(target / aspect)).long()
    ok = (w > 0) & (w <= width) & (h > 0) & (h <= height)
    first = torch.where(ok.any(1), ok.float().argmax(1), torch.zeros(batch, dtype=torch.long))
    idx = torch.arange(batch)
    w, h = w[idx, first], h[idx, first]
    fi, fj, fh, fw = random_resized_crop_params(height, width, (2.0, 2.0), ratio)     # scale 2 never fits: the fallback box
    none = ~ok.any(1)
    w = torch.where(none, torch.full_like(w, fw), w)
    h = torch.where(none, torch.full_like(h, fh), h)
    ri = torch.rand(batch, generator=generator, dtype=torch.float64)
    rj = torch.rand(batch, generator=generator, dtype=torch.float64)
    top = torch.minimum((ri * (height - h + 1).double()).long(), height - h)
    left = torch.minimum((rj * (width - w + 1).double()).long(), width - w)
    top = torch.where(none, torch.full_like(top, fi), top)
    left = torch.where(none, torch.full_like(left, fj), left)
    return torch.stack([top, left, h, w], 1).to(torch.int32).contiguous()


def sample_color_jitter(batch, brightness, contrast, saturation, hue, prob, generator=None):
    """Per-sample parameters of `color_jitter(brightness, contrast, saturation, hue, p)` (open_clip/transform.py:61-72 around
    torchvision ColorJitter.get_params): apply ~ Bernoulli(prob) uint8 [B], a random permutation of the four operations
    int32 [B,4] (0 brightness, 1 contrast, 2 saturation, 3 hue) and factors f32 [B,4] indexed by operation: U(max(0, 1 - x), 1 + x)
    for the first three, U(-hue, hue) for the hue shift."""
    apply = (torch.rand(batch, generator=generator) < prob).to(torch.uint8)
    order = torch.argsort(torch.rand(batch, 4, generator=generator), dim=1).to(torch.int32).contiguous()
    lo = torch.tensor([max(0.0, 1 - brightness), max(0.0, 1 - contrast), max(0.0, 1 - saturation), -hue])
    hi = torch.tensor([1 + brightness, 1 + contrast, 1 + saturation, hue])
    factors = (lo + (hi - lo) * torch.rand(batch, 4, generator=generator)).to(torch.float32).contiguous()
    return apply, order, factors


class DeviceAugment:
    """The reference's train transform (open_clip/transform.py:152-168) on a staged uint8 NHWC device batch:
    RandomResizedCrop(image_size, scale, ratio, BICUBIC) -> [color_jitter(b, c, s, h) with probability color_jitter_prob] ->
    [gray_scale with probability gray_scale_prob].  Returns uint8 [B, 3, S, S] in channels_last memory - the model's input
    format.  Random parameters are sampled on the host (a few bytes per sample); the pixels never leave the device."""

    def __init__(self, image_size, scale=(0.9, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), color_jitter=None, color_jitter_prob=0.0,
                 gray_scale_prob=0.0, seed=0):
        self.size, self.scale, self.ratio = int(image_size), tuple(scale), tuple(ratio)
        self.jitter = tuple(float(v) for v in color_jitter) if color_jitter else None
        if self.jitter is not None and len(self.jitter) != 4:
            raise ValueError("color_jitter = (brightness, contrast, saturation, hue)")
        self.jitter_p = float(color_jitter_prob or 0.0) if self.jitter else 0.0
        self.gray_p = float(gray_scale_prob or 0.0)
        self.gen = torch.Generator(device="cpu").manual_seed(int(seed))

    def sample(self, B, Hs, Ws):
        """-> boxes int32 [B,4], (apply, order, factors) | None, gray uint8 [B] | None: one batch's random parameters (CPU)."""
        boxes = sample_crop_boxes(B, Hs, Ws, self.scale, self.ratio, self.gen)
        jit = sample_color_jitter(B, *self.jitter, self.jitter_p, self.gen) if self.jitter_p > 0 else None
        gray = (torch.rand(B, generator=self.gen) < self.gray_p).to(torch.uint8) if self.gray_p > 0 else None
        return boxes, jit, gray

    def __call__(self, staged):
        from . import ops
        B, Hs, Ws, _ = staged.shape
        boxes, jit, gray = self.sample(B, Hs, Ws)
        put = (lambda t: t.pin_memory().to(staged.device, non_blocking=True)) if staged.is_cuda else (lambda t: t)
        boxes = put(boxes)
        gray = put(gray) if gray is not None else None
        if jit is None:                       # grayscale rides in the resize kernel's store
            return ops.resized_crop_u8(staged, boxes, self.size, gray).permute(0, 3, 1, 2)
        out = ops.resized_crop_u8(staged, boxes, self.size)
        apply, order, factors = (put(t) for t in jit)
        ops.color_jitter_u8_(out, apply, order, factors, gray)
        return out.permute(0, 3, 1, 2)


class DevicePrefetcher:
    """Iterate a host loader of (images uint8 [B,H,W,3] or [B,3,H,W], texts int64 [B,ctx]) batches `depth` batches ahead:
    each batch is staged in a pinned buffer, copied on a dedicated HIP stream and (optionally) augmented there, so the H2D
    copy of training/train.py:187-189 and the transform overlap the previous step instead of preceding this one."""

    def __init__(self, loader, device, transform=None, depth=2):
        self.loader, self.device, self.transform, self.depth = loader, torch.device(device), transform, max(1, int(depth))
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self._pins = [dict() for _ in range(self.depth)]      # slot -> {name: pinned tensor}
        self._free = [None] * self.depth                       # slot -> event after which the pinned buffers may be rewritten

    def _stage(self, slot, name, t):
        t = torch.as_tensor(t)
        if not self.cuda or t.is_pinned():
            return t
        buf = self._pins[slot].get(name)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._pins[slot][name] = buf
        buf.copy_(t)
        return buf

    def _submit(self, slot, batch):
        images, texts = batch
        if self._free[slot] is not None:
            self._free[slot].synchronize()                     # the copy that last read this slot's pinned buffers is done
        images, texts = self._stage(slot, "images", images), self._stage(slot, "texts", texts)
        if not self.cuda:
            return (self.transform(images) if self.transform else images), texts, None
        with torch.cuda.stream(self.stream):
            d_img = images.to(self.device, non_blocking=True)
            d_txt = texts.to(self.device, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(self.stream)
            self._free[slot] = copied
            if self.transform is not None:
                d_img = self.transform(d_img)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return d_img, d_txt, ready

    def __iter__(self):
        queue = collections.deque()
        slot = 0
        for batch in self.loader:
            queue.append(self._submit(slot, batch))
            slot = (slot + 1) % self.depth
            if len(queue) >= self.depth:
                yield self._hand_over(queue.popleft())
        while queue:
            yield self._hand_over(queue.popleft())

    def _hand_over(self, item):
        d_img, d_txt, ready = item
        if ready is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ready)
            d_img.record_stream(cur)
            d_txt.record_stream(cur)
        return d_img, d_txt
