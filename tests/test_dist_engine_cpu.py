"""world_size-2 gloo test (CPU) of the data-parallel training step: the real clipa_amd engine / model / loss
code under the reference's DistributedDataParallel wrapper, `ops` swapped for the torch-CPU stand-ins of
tests/cpu_ops.py (in the worker processes of this test only).  What it pins without a GPU: the engine's
autograd nodes fire DDP's gradient hooks for every parameter, the fused all-gather / reduce-scatter loss
glue, label offsets per rank, and that DDP's averaged gradient equals the global-batch gradient."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from .conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _swap():
    from clipa_amd import engine, loss as loss_mod, model as model_mod, optim as optim_mod
    from tests import cpu_ops
    for mod in (engine, loss_mod, model_mod, optim_mod):
        mod.ops = cpu_ops


def _model(g):
    import clipa_amd
    m = clipa_amd.CLIP(**g.cfg, output_dict=True)
    m.load_state_dict(g.sd, strict=True)
    m.set_grad_checkpointing(True)
    m.visual.transformer.keep_blocks, m.visual.transformer.medium_blocks = 0, 1      # mixed activation tiers
    return m


def _worker(rank, world, port, q, unpad=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import clipa_amd
    _swap()
    g = load_golden("cls_erf")
    m = _model(g)
    m.unpad_text = bool(unpad)      # before the wrap: DDP(static_graph=True) records the autograd graph of its first iteration
    ddp = torch.nn.parallel.DistributedDataParallel(m, static_graph=True)
    B = g.images_u8.shape[0] // world
    img, txt = g.images_u8[rank * B:(rank + 1) * B], g.texts[rank * B:(rank + 1) * B]
    lens = (txt.argmax(-1) + 1).tolist() if unpad else None      # from the loader's host-side token ids, as a LIST: DDP moves tensor kwargs to the device
    loss_fn = clipa_amd.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True, rank=rank, world_size=world).bind(ddp)
    losses, hits = [], []
    for _ in range(4):
        ddp.zero_grad(set_to_none=True)
        out = ddp(img, txt, text_lengths=lens)
        loss = loss_fn(**out, output_dict=True)["contrastive_loss"]
        loss.backward()
        losses.append(float(loss.detach()))
        hits.append(loss_fn.early_hits)
    # DDP(static_graph=True) clones the outputs on its first iteration (one unused early gather); from the second
    # iteration on the overlapped image-feature gather must really be the one the loss consumes
    assert loss_fn._early_ok and hits[-1] - hits[0] == 3, hits
    grads = {n: p.grad.detach().float().numpy() for n, p in m.named_parameters() if p.grad is not None}
    q.put((rank, losses, grads))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("unpad", [False, True])
def test_two_rank_ddp_step_equals_global_batch_step(unpad):
    """unpad: every rank runs its text tower on the tokens up to EOT (`unpad_text`, set before the DDP wrap; ragged packed
    matrices that differ between the ranks) - the step still equals the padded single-process global-batch step."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, 29757 + int(unpad), q, unpad)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, losses, grads = q.get(timeout=300)
        got[rank] = (losses, grads)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    import clipa_amd
    _swap()
    try:
        g = load_golden("cls_erf")
        m = _model(g)
        out = m(g.images_u8, g.texts)
        loss = clipa_amd.ClipLoss()(**out, output_dict=True)["contrastive_loss"]
        loss.backward()
        ref = {n: p.grad.detach().float().numpy() for n, p in m.named_parameters() if p.grad is not None}
    finally:
        import importlib
        from clipa_amd import engine, loss as loss_mod, model as model_mod, optim as optim_mod, ops as real_ops
        for mod in (engine, loss_mod, model_mod, optim_mod):
            mod.ops = real_ops
    mean_local = 0.5 * (got[0][0][0] + got[1][0][0])
    assert abs(mean_local - float(loss)) <= 2e-3 * abs(float(loss)), (mean_local, float(loss))
    assert max(abs(got[0][0][0] - l) for l in got[0][0][1:]) < 1e-6
    assert set(got[0][1]) == set(ref)
    for n, gref in ref.items():
        a, b = got[0][1][n], got[1][1][n]
        assert np.array_equal(a, b), f"ranks disagree on {n} after the all-reduce"
        cos = float((a * gref).sum() / (np.linalg.norm(a) * np.linalg.norm(gref) + 1e-30))
        rel = float(np.linalg.norm(a) / (np.linalg.norm(gref) + 1e-30))
        assert cos >= 0.995 and 0.97 <= rel <= 1.03, (n, cos, rel)
