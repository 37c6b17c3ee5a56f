"""Contrastive (InfoNCE) loss with cross-GPU feature gather, computed by the HIP engine.

Drop-in for `open_clip.loss.ClipLoss` (clipa_torch/open_clip/loss.py:92-157) and its
`gather_features` torch.distributed branch (loss.py:73-87): same constructor, same call signature
`loss(image_features, text_features, logit_scale, output_dict=False)`, same variants
(local_loss x gather_with_grad), same label convention (arange + B*rank, loss.py:115-126).

MI355X mapping.
* Gather: each rank's embeddings travel as bf16 over RCCL (torch.distributed backend "nccl" is RCCL on ROCm; xGMI
  underneath) on a side HIP stream.  The IMAGE embeddings can be gathered EARLY: after `loss.bind(model)` (explicit
  opt-in, one line next to create_loss) `CLIP.forward` hands them to `ClipLoss.early_gather()` as soon as the image
  tower's projection + normalisation are done, so the all-gather runs concurrently with the whole text tower on the
  compute stream; the loss picks the result up (`self._pending`, matched by tensor IDENTITY) and only the text gather
  is exposed.  (The reference gathers both serially after both towers, loss.py:73-76.)  All of that state lives on the
  ClipLoss instance; a forward whose features never reach the loss unchanged (accum_freq > 1 concatenates them,
  train.py:242-247; the first iteration of DistributedDataParallel(static_graph=True) clones its outputs) costs one unused
  gather; after three consecutive ones the instance stops gathering early.
* Backward of the differentiable gather = ONE reduce-scatter(SUM) of the fused [W*B, 2E] fp32 gradient - the
  semantics of torch.distributed.nn.all_gather's backward used by the reference.
* Similarity GEMM and cross-entropy are ONE kernel pair (`ops.simce`, csrc/simce.hip): the GEMM epilogue reduces each
  logits tile to per-row (max, sum exp) partials, the fp32 [B, W*B] logits never reach HBM; the backward re-runs the
  GEMM and emits the bf16 d loss / d (I.T^T) that the four gradient GEMMs consume.  exp(logit_scale) stays in device
  memory and is applied inside those kernels - no `.item()`, no host sync per step.  Batches that are not multiples
  of 8 are zero-padded to the GEMM granularity; the pad columns are excluded from the softmax inside the kernel.
* Host-side glue that stays in torch: `torch.cat` / `+` of the [B, E] gradient pieces in backward (a few MB).
"""
import weakref

import torch
import torch.distributed as dist
from torch import nn

from . import ops

bf16, f32 = torch.bfloat16, torch.float32

_SIDE = {}                                    # device -> side stream (created once per device, never mutated afterwards)
_WAITS = None                                 # bench instrumentation: [(event before, event after)] around the gather waits


def wait_timing_start():
    """bench.py: record a HIP event pair around every wait of the compute stream for a feature gather (the EXPOSED part of
    the exchange: whatever the side stream has not finished by the time the loss needs the gathered rows)."""
    global _WAITS
    _WAITS = []


def wait_timing_stop():
    """-> total milliseconds the compute stream spent waiting for gathers since wait_timing_start()."""
    global _WAITS
    recs, _WAITS = _WAITS, None
    if not recs:
        return 0.0
    torch.cuda.synchronize()
    return float(sum(a.elapsed_time(b) for a, b in recs))



def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def _all_gather_bf16(local, world_size, group=None):
    """all-gather [B, E] bf16 -> [W*B, E], issued on the side stream; returns (gathered, event | None).  The caller's
    stream must wait for the event before it reads `gathered`."""
    out = torch.empty((world_size * local.shape[0], local.shape[1]), device=local.device, dtype=local.dtype)
    if local.is_cuda:
        cur = torch.cuda.current_stream()
        side = _side_stream(local.device)
        side.wait_stream(cur)                         # `local` was produced on the compute stream
        with torch.cuda.stream(side):
            dist.all_gather_into_tensor(out.view(torch.uint8), local.view(torch.uint8), group=group)   # pure byte movement
            ev = torch.cuda.Event()
            ev.record(side)
        local.record_stream(side)
        out.record_stream(side)
        return out, ev
    dist.all_gather_into_tensor(out, local, group=group)
    return out, None


def _reduce_scatter_fused(full, world_size, group=None):
    """reduce-scatter(SUM) [W*B, 2E] f32 -> [B, 2E]: backward of the differentiable all-gather."""
    rows = full.shape[0] // world_size
    if dist.get_backend(group) == "gloo" and full.is_cuda:   # 2-ranks-on-one-GPU test harness only: gloo reduce-scatters CPU
        # tensors (the 2- and 8-rank CPU tests walk the production call below), not device tensors: all-reduce, keep our slice
        full = full.contiguous().clone()
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
        r = dist.get_rank(group)
        return full[r * rows:(r + 1) * rows]
    out = torch.empty((rows, full.shape[1]), device=full.device, dtype=full.dtype)
    dist.reduce_scatter_tensor(out, full.contiguous(), op=dist.ReduceOp.SUM, group=group)
    return out


def _pad_rows8(x):
    """[n, E] -> [n8, E] with zero rows (the similarity GEMMs need N, and their gradients K, in multiples of 8)."""
    n = x.shape[0]
    n8 = (n + 7) // 8 * 8
    if n8 == n:
        return x
    out = torch.zeros((n8, x.shape[1]), device=x.device, dtype=x.dtype)
    out[:n] = x
    return out


class ClipLossFn(torch.autograd.Function):
    """(CE(logits_per_image, y) + CE(logits_per_text, y)) / 2 with logits = s * I . T^T."""

    @staticmethod
    def forward(ctx, img, txt, logit_scale, local_loss, gather_with_grad, rank, world_size, group, pend=None):
        B, E = img.shape
        s_dev = logit_scale.detach().to(f32).reshape(1).contiguous()     # stays on the device
        tb = ops.to_bf16(txt)
        if world_size > 1:
            t_all, t_ev = _all_gather_bf16(tb, world_size, group)
            if pend is not None:
                ib, i_all, i_ev = pend                                    # started before the text tower ran
            else:
                ib = ops.to_bf16(img)
                i_all, i_ev = _all_gather_bf16(ib, world_size, group)
            if img.is_cuda:
                cur = torch.cuda.current_stream()
                if _WAITS is not None:
                    before = torch.cuda.Event(enable_timing=True)
                    before.record(cur)
                for ev in (i_ev, t_ev):
                    if ev is not None:
                        cur.wait_event(ev)
                if _WAITS is not None:
                    after = torch.cuda.Event(enable_timing=True)
                    after.record(cur)
                    _WAITS.append((before, after))
        else:
            ib = ops.to_bf16(img)
            i_all, t_all = ib, tb
        if world_size > 1 and not local_loss:
            i_rows, t_rows, label0 = i_all, t_all, 0
        else:
            i_rows, t_rows, label0 = ib, tb, (B * rank if world_size > 1 else 0)
        R, N = i_rows.shape[0], i_all.shape[0]
        i_all8, t_all8 = _pad_rows8(i_all), _pad_rows8(t_all)
        need_grad = any(ctx.needs_input_grad[:3])
        gs = 0.5 / R
        # fused similarity GEMM + cross-entropy: the [R, N] fp32 logits never reach HBM (K17 + K18)
        li, dli, dsi = ops.simce(i_rows, t_all8, N, label0, gs, scale=s_dev, want_grad=need_grad)
        lt, dlt, dst = ops.simce(t_rows, i_all8, N, label0, gs, scale=s_dev, want_grad=need_grad)
        loss = ops.sum_scale(li, gs)
        ops.sum_scale(lt, gs, out=loss, accumulate=True)
        if need_grad:
            ctx.save_for_backward(dli, dlt, dsi, dst, i_rows, t_rows, i_all8, t_all8)
            ctx.meta = (B, E, N, local_loss, gather_with_grad, rank, world_size, group, img.dtype, txt.dtype,
                        logit_scale.dtype, logit_scale.shape)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        dli, dlt, dsi, dst, i_rows, t_rows, i_all8, t_all8 = ctx.saved_tensors
        B, E, N, local_loss, gather_with_grad, rank, W, group, idt, tdt, sdt, sshape = ctx.meta
        # dli / dlt = d loss / d raw (the scale is already folded in).  raw_i = I_rows T_all^T ; raw_t = T_rows I_all^T
        d_i_rows = ops.gemm_nt(dli, ops.transpose_bf16(t_all8), out_f32=True)              # [R, E]
        d_t_rows = ops.gemm_nt(dlt, ops.transpose_bf16(i_all8), out_f32=True)
        # d/d(gathered columns)
        d_t_all = ops.gemm_tn(dli, i_rows, f32)[:N]                                        # [W*B, E]
        d_i_all = ops.gemm_tn(dlt, t_rows, f32)[:N]
        if W == 1:
            d_i = d_i_rows + d_i_all
            d_t = d_t_rows + d_t_all
        elif local_loss:
            d_i, d_t = d_i_rows, d_t_rows
            if gather_with_grad:
                rs = _reduce_scatter_fused(torch.cat([d_i_all, d_t_all], dim=1), W, group)
                d_i = d_i + rs[:, :E]
                d_t = d_t + rs[:, E:]
        else:
            # rows are the gathered features themselves: every term is a gradient w.r.t. the gather
            full = torch.cat([d_i_rows + d_i_all, d_t_rows + d_t_all], dim=1)
            if gather_with_grad:
                rs = _reduce_scatter_fused(full, W, group)
            else:
                rs = full[rank * B:(rank + 1) * B]      # only the re-inserted local slice carries grad
            d_i, d_t = rs[:, :E], rs[:, E:]
        d_s = ops.sum_scale(dsi, 1.0)
        ops.sum_scale(dst, 1.0, out=d_s, accumulate=True)
        g = dloss.to(f32)
        return ((d_i * g).to(idt), (d_t * g).to(tdt), (d_s * g).to(sdt).reshape(sshape), None, None, None, None, None,
                None)


class ClipLoss(nn.Module):
    """Same constructor / call contract as open_clip.loss.ClipLoss (loss.py:92-157)."""

    def __init__(self, local_loss=False, gather_with_grad=False, cache_labels=False, rank=0, world_size=1,
                 use_horovod=False, group=None):
        super().__init__()
        if use_horovod:
            raise NotImplementedError("clipa_amd: horovod is outside the MI355X path (RCCL via torch.distributed)")
        self.local_loss = local_loss
        self.gather_with_grad = gather_with_grad
        self.cache_labels = cache_labels     # labels are implicit (arange + offset) in the CE kernel
        self.rank = rank
        self.world_size = world_size
        self.use_horovod = use_horovod
        self.group = group
        self._pending = None        # (weakref(features), version, local bf16, gathered bf16, side-stream event)
        self._early_ok = True       # cleared when EARLY_MISS_LIMIT consecutive early gathers went unused
        self._early_misses = 0      # consecutive early gathers whose tensor never reached the loss unchanged
        self.early_hits = 0         # forwards that consumed an early gather (diagnostics / tests)

    # An early gather goes unused when the loss does not see the very tensor the model returned: accum_freq > 1 (the loss
    # sees torch.cat of cached features, train.py:242-247) misses on EVERY step, DistributedDataParallel(static_graph=True)
    # only on its first iteration (its _DDPSink clones the outputs once).  One miss must therefore not switch the overlap
    # off for good: it is switched off after this many CONSECUTIVE misses, and a hit re-arms the count.  Every rank runs
    # the same forwards, so every rank takes the same decision at the same step (early_gather is a collective).
    EARLY_MISS_LIMIT = 3

    def _early_miss(self):
        self._early_misses += 1
        if self._early_misses >= self.EARLY_MISS_LIMIT:
            self._early_ok = False

    def bind(self, model):
        """Opt in to the overlapped image-feature gather: `model` (a clipa_amd.CLIP, possibly inside DDP) will call
        `self.early_gather` from its forward.  EVERY rank must then run the same grad-enabled training forwards (the
        call is a collective), exactly as every rank must call the loss.  Returns self."""
        getattr(model, "module", model)._gather_partner = weakref.ref(self)
        return self

    def early_gather(self, features):
        """Start the all-gather of the (normalised, f32 [B, E]) image features on the side stream so that it overlaps
        the text tower.  No-op for a single rank or once an early gather has gone unused."""
        if self.world_size <= 1 or not self._early_ok or not dist.is_available() or not dist.is_initialized():
            return
        if self._pending is not None:          # the previous forward never reached the loss at all (e.g. a no-loss forward)
            self._pending = None
            self._early_miss()
            if not self._early_ok:
                return
        local = ops.to_bf16(features.detach())
        gathered, ev = _all_gather_bf16(local, self.world_size, self.group)
        self._pending = (weakref.ref(features), features._version, local, gathered, ev)

    def _take_pending(self, features):
        ent, self._pending = self._pending, None
        if ent is None:
            return None
        if ent[0]() is features and ent[1] == features._version:
            self._early_misses = 0
            self.early_hits += 1
            return ent[2:]
        self._early_miss()                     # e.g. accum_freq > 1: the loss sees torch.cat(...) of cached features
        return None

    def forward(self, image_features, text_features, logit_scale, output_dict=False):
        pend = self._take_pending(image_features) if self.world_size > 1 else None
        # the reference trainer calls the loss inside torch.autocast (train.py:203-213); the kernels take fixed dtypes
        with torch.autocast(device_type=image_features.device.type, enabled=False):
            if not torch.is_tensor(logit_scale):
                logit_scale = torch.tensor(float(logit_scale), device=image_features.device)
            total_loss = ClipLossFn.apply(image_features.float(), text_features.float(), logit_scale, self.local_loss,
                                          self.gather_with_grad, self.rank, self.world_size, self.group, pend)
        return {"contrastive_loss": total_loss} if output_dict else total_loss
