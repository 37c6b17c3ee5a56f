"""CPU tests of clipa_amd.zero.ShardedAdamW: the sharded gradient exchange + sharded optimizer state that
replaces DDP's all-reduce + a full AdamW per rank (clipa_torch/training/main.py:292-299,318-326).  `ops` is swapped for
the torch-CPU stand-ins (tests/cpu_ops.py); transport = gloo, world_size 2.
  * one process: flat-buffer AdamW == clipa_amd.optim.AdamW (same kernel maths), parameters stay views of the flat buffer;
  * two ranks: after one step every rank holds the SAME parameters, equal to a single AdamW step on the rank-averaged
    gradient (DDP semantics) - for a plain torch model and for the CLIP engine with its all-gather loss;
  * no_sync accumulation, state_dict round trip into torch.optim.AdamW's layout."""
import math
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from .conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HP = dict(lr=1e-2, betas=(0.9, 0.95), eps=1e-6)


def _swap():
    from clipa_amd import engine, loss as loss_mod, model as model_mod, optim as optim_mod, zero as zero_mod
    from tests import cpu_ops
    for mod in (engine, loss_mod, model_mod, optim_mod, zero_mod):
        mod.ops = cpu_ops


def _unswap():
    from clipa_amd import engine, loss as loss_mod, model as model_mod, optim as optim_mod, zero as zero_mod, ops as real_ops
    for mod in (engine, loss_mod, model_mod, optim_mod, zero_mod):
        mod.ops = real_ops


@pytest.fixture
def cpu_ops_swapped():
    _swap()
    yield
    _unswap()


def _toy(seed=0, dtype=torch.float32):
    torch.manual_seed(seed)
    m = torch.nn.Sequential(torch.nn.Linear(24, 40), torch.nn.GELU(), torch.nn.Linear(40, 16), torch.nn.LayerNorm(16),
                            torch.nn.Linear(16, 8))
    return m.to(dtype)


def _groups(m):
    decay = [p for p in m.parameters() if p.ndim >= 2]
    rest = [p for p in m.parameters() if p.ndim < 2]
    return [{"params": rest, "weight_decay": 0.0}, {"params": decay, "weight_decay": 0.2}]


def test_single_process_equals_fused_adamw(cpu_ops_swapped):
    from clipa_amd.optim import AdamW
    from clipa_amd.zero import ShardedAdamW
    a, b = _toy(), _toy()
    oa = AdamW(_groups(a), grad_clip_norm=0.5, **HP)
    ob = ShardedAdamW(_groups(b), grad_clip_norm=0.5, bucket_bytes=1024, **HP)
    assert len(ob.buckets) > 2                                        # several buckets per group at this bucket size
    for p in b.parameters():
        assert any(bk.flat.data_ptr() <= p.data_ptr() < bk.flat.data_ptr() + bk.flat.numel() * 4 for bk in ob.buckets)
    x = torch.randn(32, 24)
    for step in range(3):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad()
            (m(x) ** 2).mean().backward()
            o.step()
        for p, q in zip(a.parameters(), b.parameters()):
            assert torch.equal(p, q), step
    assert abs(float(oa.last_grad_norm) - float(ob.last_grad_norm)) < 1e-6
    sd = ob.state_dict()                                              # torch.optim.AdamW's layout: loads into the reference's optimizer
    ref = torch.optim.AdamW(_groups(_toy()), **HP)
    ref.load_state_dict(sd)
    st = oa.state_dict()["state"]
    for k, v in sd["state"].items():
        assert torch.equal(v["exp_avg"], st[k]["exp_avg"]) and torch.equal(v["exp_avg_sq"], st[k]["exp_avg_sq"])
    c = _toy()
    oc = ShardedAdamW(_groups(c), grad_clip_norm=0.5, bucket_bytes=1 << 20, **HP)   # a different bucket layout
    c.load_state_dict(b.state_dict())
    oc.load_state_dict(sd)
    for m, o in ((b, ob), (c, oc)):
        o.zero_grad()
        (m(x) ** 2).mean().backward()
        o.step()
    for p, q in zip(b.parameters(), c.parameters()):
        assert torch.equal(p, q)


def _toy_worker(rank, world, port, q, exchange, mode="no_sync", tensor_collectives=None):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _swap()
    from clipa_amd.zero import ShardedAdamW
    m = _toy(seed=rank)                                                # different initial weights: rank 0's must win
    opt = ShardedAdamW(_groups(m), grad_clip_norm=0.5, bucket_bytes=1024, exchange=exchange, tensor_collectives=tensor_collectives,
                       **HP)
    torch.manual_seed(100 + rank)
    xs = [torch.randn(16, 24) for _ in range(3)]
    for step in range(2):
        opt.zero_grad()
        if mode == "no_sync":
            with opt.no_sync():                                        # two micro-batches accumulate, the third exchanges
                (m(xs[0]) ** 2).mean().backward()
                (m(xs[1]) ** 2).mean().backward()
            (m(xs[2]) ** 2).mean().backward()
        elif mode == "plain":                                          # the reference's accum loop (train.py:243-256): backward()
            for x in xs:                                               # accum_freq times, NO no_sync - every bucket re-exchanges
                (m(x) ** 2).mean().backward()
        else:                                                          # sync backward first, no_sync afterwards: step() must catch up
            (m(xs[0]) ** 2).mean().backward()
            with opt.no_sync():
                (m(xs[1]) ** 2).mean().backward()
                (m(xs[2]) ** 2).mean().backward()
        opt.step()
    sd = opt.state_dict()
    q.put((rank, [p.detach().numpy().copy() for p in m.parameters()], float(opt.last_grad_norm),
           {k: v["exp_avg"].numpy().copy() for k, v in sd["state"].items()}))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(target, world, port, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=300)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("mode,port,world,exchange,tc", [
    ("no_sync", 29771, 2, "reduce_scatter", None), ("plain", 29775, 2, "reduce_scatter", None),
    ("sync_then_no_sync", 29777, 2, "reduce_scatter", None),
    # the node size of the BASELINE configurations, on the PRODUCTION collectives (gloo has them for CPU tensors): buckets of
    # 1 KiB whose padded sizes are multiples of 8 x 8 elements - i.e. the tail-handling RCCL's AVG path got wrong in round 2
    ("plain", 29779, 8, "reduce_scatter", True), ("no_sync", 29781, 8, "all_to_all", True),
    ("sync_then_no_sync", 29783, 8, "all_to_all", True)])
def test_sharded_step_equals_adamw_on_averaged_gradient(mode, port, world, exchange, tc):
    """Three ways of accumulating three micro-batches before one step - inside no_sync (one exchange), plain repeated
    backward() as the reference's accum_freq loop does (every completed round re-exchanges the cumulative gradient), and
    an exchange that later backwards make stale - all equal AdamW on the rank-averaged accumulated gradient; with 2 ranks
    (gloo's all-reduce emulation, as the 2-ranks-on-one-GPU tests use) and with 8 ranks on reduce_scatter_tensor /
    all_to_all_single + reduce_shards / all_gather_into_tensor."""
    got = _spawn(_toy_worker, world, port, exchange, mode, tc)
    _swap()
    try:
        from clipa_amd.optim import AdamW
        m = _toy(seed=0)
        opt = AdamW(_groups(m), grad_clip_norm=0.5, **HP)
        for step in range(2):
            opt.zero_grad()
            for rank in range(world):                                  # the rank-averaged gradient of the accumulated micro-batches
                torch.manual_seed(100 + rank)
                xs = [torch.randn(16, 24) for _ in range(3)]
                for x in xs:
                    ((m(x) ** 2).mean() / world).backward()
            opt.step()
        ref = [p.detach().numpy() for p in m.parameters()]
        ref_m = {k: v["exp_avg"].numpy() for k, v in opt.state_dict()["state"].items()}
    finally:
        _unswap()
    for k in range(1, world):
        for a, b in zip(got[0][0], got[k][0]):
            assert np.array_equal(a, b), f"ranks 0 and {k} hold different parameters after the all-gather"
        assert abs(got[0][1] - got[k][1]) < 1e-6
    for a, r in zip(got[0][0], ref):
        assert np.allclose(a, r, rtol=2e-5, atol=2e-6)
    for k, v in ref_m.items():
        assert np.allclose(got[0][2][k], v, rtol=2e-5, atol=1e-7), k   # consolidated moments = the unsharded optimizer's


def _clip_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import clipa_amd
    _swap()
    from clipa_amd.zero import ShardedAdamW
    g = load_golden("cls_erf")
    m = clipa_amd.CLIP(**g.cfg, output_dict=True)
    m.load_state_dict(g.sd, strict=True)
    m.set_grad_checkpointing(True)
    opt = ShardedAdamW([p for p in m.parameters() if p.requires_grad], weight_decay=0.0, bucket_bytes=64 << 10,
                       clamp=(m.logit_scale, 0.0, math.log(100)), **dict(HP, lr=2e-3))
    B = g.images_u8.shape[0] // world
    img, txt = g.images_u8[rank * B:(rank + 1) * B], g.texts[rank * B:(rank + 1) * B]
    loss_fn = clipa_amd.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True, rank=rank, world_size=world).bind(m)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        out = m(img, txt)                                              # no DDP wrapper: the optimizer's hooks do the exchange
        loss = loss_fn(**out, output_dict=True)["contrastive_loss"]
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    q.put((rank, losses, {n: p.detach().float().numpy().copy() for n, p in m.named_parameters()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_clip_training_with_sharded_optimizer():
    world = 2
    got = _spawn(_clip_worker, world, 29773)
    import clipa_amd
    _swap()
    try:
        from clipa_amd.optim import AdamW
        g = load_golden("cls_erf")
        m = clipa_amd.CLIP(**g.cfg, output_dict=True)
        m.load_state_dict(g.sd, strict=True)
        m.set_grad_checkpointing(True)
        opt = AdamW([p for p in m.parameters() if p.requires_grad], weight_decay=0.0,
                    clamp=(m.logit_scale, 0.0, math.log(100)), **dict(HP, lr=2e-3))
        ref_losses = []
        for _ in range(3):
            opt.zero_grad()
            out = m(g.images_u8, g.texts)
            loss = clipa_amd.ClipLoss()(**out, output_dict=True)["contrastive_loss"]
            loss.backward()
            opt.step()
            ref_losses.append(float(loss.detach()))
        ref = {n: p.detach().float().numpy() for n, p in m.named_parameters()}
    finally:
        _unswap()
    for n in ref:
        assert np.array_equal(got[0][1][n], got[1][1][n]), n            # every rank ends with the same weights
    mean_losses = [0.5 * (a + b) for a, b in zip(got[0][0], got[1][0])]
    for a, b in zip(mean_losses, ref_losses):                           # the 2-rank run follows the global-batch run
        assert abs(a - b) < 5e-3 * abs(b) + 1e-3, (mean_losses, ref_losses)
    assert ref_losses[-1] < ref_losses[0] and mean_losses[-1] < mean_losses[0]
    worst = 0.0
    for n, r in ref.items():
        d = np.abs(got[0][1][n] - r).max()
        worst = max(worst, float(d))
    assert worst < 1.5e-2, worst        # AdamW normalises the step: a sign flip of a ~0 gradient moves a weight by 2 * lr per step
