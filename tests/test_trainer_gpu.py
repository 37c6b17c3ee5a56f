"""GPU: the reference's training step loop driving the HIP engine under `torch.autocast(bfloat16)`.

Every reference GPU script trains with `--precision amp_bf16` (clipa_torch/scripts/exp/gpu/vit_l16/i37_t8_pretrain.sh:18):
`train_one_epoch` wraps `model(...)` and `loss(...)` in `torch.cuda.amp.autocast(dtype=torch.bfloat16)`
(training/train.py:160,203-213; training/precision.py:6-14).  /root/reference does not exist on the GPU box, so the loop
here is `oracle.trainer_harness.restated_train_one_epoch` - held bit for bit to the real `train_one_epoch` by
tests/test_trainer_cpu.py in the build container - and the expected numbers are tests/golden/train_loop_ref.npz, produced
by the REAL trainer driving the REAL reference CLIP (oracle/make_train_golden.py): per-call loss (<= 2 %), clamped
logit_scale, per-parameter update size, and the update direction of nine parameters (cosine >= 0.97).
"""
import numpy as np
import pytest
import torch

import clipa_amd
from oracle import clip_oracle as O
from oracle import trainer_harness as H
from oracle.make_golden import CASES

pytestmark = pytest.mark.gpu
DEV = "cuda"
Z = np.load("tests/golden/train_loop_ref.npz")


def _autocast():
    return torch.autocast("cuda", dtype=torch.bfloat16)


def _run(accum, nsteps, opt_cls=torch.optim.AdamW, autocast=_autocast):
    spec = CASES["cls_erf"]
    cfg, B, S, seed = spec["cfg"], spec["B"], spec["S"], spec["seed"]
    m = clipa_amd.CLIP(**cfg, output_dict=True)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd0 = O.make_state_dict(shapes, seed)
    m.load_state_dict(sd0, strict=True)
    m.visual.image_mean, m.visual.image_std = H.IMAGE_MEAN, H.IMAGE_STD
    m.to(DEV)
    m.set_grad_checkpointing(True)
    tc = cfg["text_cfg"]
    batches = H.synthetic_batches(nsteps * accum, B, S, tc["context_length"], tc["vocab_size"], seed + 100)
    args = H.make_args(DEV, batch_size=B, accum_freq=accum)
    opt = H.make_optimizer(m, opt_cls)
    loss = H.LossRecorder(clipa_amd.ClipLoss(local_loss=False, gather_with_grad=False, cache_labels=True, rank=0, world_size=1))
    H.restated_train_one_epoch(m, batches, loss, opt, H.make_scheduler(opt), args, autocast)
    torch.cuda.synchronize()
    return m, sd0, loss.values


def _check(tag, m, sd0, losses, cos_min=0.97):
    ref = Z[f"{tag}_losses"]
    assert len(losses) == len(ref)
    for a, b in zip(losses, ref):
        assert abs(a - b) < 2e-2 * b, (losses, list(ref))
    sd1 = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    assert abs(float(sd1["logit_scale"]) - float(Z[f"{tag}_w::logit_scale"])) < 2e-3
    for n, dn in zip([str(n) for n in Z[f"{tag}_names"]], Z[f"{tag}_delta_norms"]):
        d = float((sd1[n].double() - sd0[n].double()).norm())
        assert abs(d - dn) <= 0.25 * dn + 1e-6, (n, d, dn)
    worst = (1.0, None)
    for k in Z["keep"]:
        k = str(k)
        if k == "logit_scale":
            continue
        da = (sd1[k].double() - sd0[k].double()).reshape(-1)
        db = (torch.from_numpy(Z[f"{tag}_w::{k}"]).double() - sd0[k].double()).reshape(-1)
        cos = float(torch.dot(da, db) / (da.norm() * db.norm()))
        worst = min(worst, (cos, k))
        assert cos > cos_min, (k, cos)
    print(f"[trainer {tag}] losses {np.round(losses, 4)} vs {np.round(ref, 4)}; worst update cosine {worst}")


@pytest.mark.parametrize("tag,accum,nsteps", [("a1", 1, 3), ("a2", 2, 2)])
def test_reference_step_loop_under_autocast_matches_reference_trainer(tag, accum, nsteps):
    m, sd0, losses = _run(accum, nsteps)
    _check(tag, m, sd0, losses)


def test_autocast_does_not_change_the_engine():
    """amp_bf16 (autocast on) and fp32-precision (no autocast) runs of the engine are the SAME computation: every FLOP is a
    HIP kernel with fixed operand types, autocast only sees the glue."""
    from contextlib import suppress
    ma, _, la = _run(1, 2)
    mb, _, lb = _run(1, 2, autocast=suppress)
    assert la == lb
    for (k, a), (_, b) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert torch.equal(a, b), k


def test_fused_adamw_in_the_reference_loop():
    """The engine's own optimizer (clipa_amd.optim.AdamW) in place of torch.optim.AdamW at main.py:318."""
    from clipa_amd.optim import AdamW
    m, sd0, losses = _run(1, 3, opt_cls=AdamW)
    _check("a1", m, sd0, losses)
