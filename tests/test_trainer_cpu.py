"""The reference's OWN trainer drives the engine (INTEGRATION.md section 1 applied for real).

`training.train.train_one_epoch` (clipa_torch/training/train.py:158-314) is imported from /root/reference with `open_clip`
bound to `clipa_amd` (oracle/trainer_harness.py) and run, unmodified, on the engine with the torch-CPU stand-in ops:
`--precision amp_bf16` with a REAL active autocast context (the CPU flavour of train.py:160's `get_autocast`, since
`torch.cuda.amp.autocast` switches itself off without CUDA), `--to-float-on-device`, clip_grad_norm_, AdamW groups by name,
accum_freq 1 and 2.  Compared with tests/golden/train_loop_ref.npz = the same trainer driving the reference's own CLIP (fp32).
Also pins the restated loop that tests/test_trainer_gpu.py uses on the GPU box (no /root/reference there) bit for bit.
"""
import numpy as np
import pytest
import torch

import clipa_amd
from clipa_amd import engine, loss as loss_mod, model as model_mod, optim as optim_mod
from oracle import clip_oracle as O
from oracle import ref_loader
from oracle import trainer_harness as H
from oracle.make_golden import CASES

from . import cpu_ops

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not mounted")
Z = np.load("tests/golden/train_loop_ref.npz")


@pytest.fixture(autouse=True)
def _swap_ops(monkeypatch):
    for mod in (engine, loss_mod, model_mod, optim_mod):
        monkeypatch.setattr(mod, "ops", cpu_ops)
    yield
    H.unload()


def _cpu_autocast():
    return torch.autocast("cpu", dtype=torch.bfloat16)


def _engine_and_data(accum, nsteps):
    spec = CASES["cls_erf"]
    cfg, B, S, seed = spec["cfg"], spec["B"], spec["S"], spec["seed"]
    m = clipa_amd.CLIP(**cfg, output_dict=True)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd0 = O.make_state_dict(shapes, seed)
    m.load_state_dict(sd0, strict=True)
    m.visual.image_mean, m.visual.image_std = H.IMAGE_MEAN, H.IMAGE_STD
    tc = cfg["text_cfg"]
    batches = H.synthetic_batches(nsteps * accum, B, S, tc["context_length"], tc["vocab_size"], seed + 100)
    return m, sd0, batches, H.make_args("cpu", batch_size=B, accum_freq=accum)


def _run_reference_trainer(accum, nsteps):
    train = H.load_trainer("engine")
    import open_clip
    assert open_clip.CLIP is clipa_amd.CLIP and train.CLIP is clipa_amd.CLIP
    train.get_autocast = lambda precision: _cpu_autocast          # an ACTIVE bf16 autocast around model(...) and loss(...)
    m, sd0, batches, args = _engine_and_data(accum, nsteps)
    opt = H.make_optimizer(m)
    sched = H.make_scheduler(opt)
    loss = H.LossRecorder(clipa_amd.ClipLoss(local_loss=False, gather_with_grad=False, cache_labels=True, rank=0, world_size=1))
    train.train_one_epoch(m, {"train": H.DataInfo(H.ListLoader(batches))}, loss, 0, opt, None, sched, None, args)
    return m, sd0, loss.values


@pytest.mark.parametrize("tag,accum,nsteps", [("a1", 1, 3), ("a2", 2, 2)])
def test_reference_train_one_epoch_drives_the_engine(tag, accum, nsteps):
    m, sd0, losses = _run_reference_trainer(accum, nsteps)
    ref = Z[f"{tag}_losses"]
    assert len(losses) == len(ref)
    for a, b in zip(losses, ref):
        assert abs(a - b) < 2e-2 * b, (losses, ref)
    sd1 = m.state_dict()
    assert abs(float(sd1["logit_scale"]) - float(Z[f"{tag}_w::logit_scale"])) < 2e-3
    names = [str(n) for n in Z[f"{tag}_names"]]
    for n, dn in zip(names, Z[f"{tag}_delta_norms"]):
        d = float((sd1[n].double() - sd0[n].double()).norm())
        assert abs(d - dn) <= 0.25 * dn + 1e-6, (n, d, dn)                      # same update size (Adam: ~lr per element)
    for k in Z["keep"]:
        k = str(k)
        if k == "logit_scale" or f"{tag}_w::{k}" not in Z:
            continue
        da = (sd1[k].double() - sd0[k].double()).reshape(-1)
        db = (torch.from_numpy(Z[f"{tag}_w::{k}"]).double() - sd0[k].double()).reshape(-1)
        cos = float(torch.dot(da, db) / (da.norm() * db.norm()))
        assert cos > 0.98, (k, cos)         # the update DIRECTION (Adam's first steps are ~ lr * sign(g): bf16 flips small entries)


@pytest.mark.parametrize("accum,nsteps", [(1, 2), (2, 1)])
def test_restated_loop_equals_reference_loop(accum, nsteps):
    """The loop tests/test_trainer_gpu.py runs on the GPU box is the reference's, operation for operation."""
    m, _, losses = _run_reference_trainer(accum, nsteps)
    m2, _, batches, args = _engine_and_data(accum, nsteps)
    opt = H.make_optimizer(m2)
    loss = H.LossRecorder(clipa_amd.ClipLoss(local_loss=False, gather_with_grad=False, cache_labels=True, rank=0, world_size=1))
    H.restated_train_one_epoch(m2, batches, loss, opt, H.make_scheduler(opt), args, _cpu_autocast)
    assert loss.values == losses
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
