"""world_size-2 and world_size-8 gloo tests (CPU) of the multi-rank glue of clipa_amd.loss: bf16 feature all-gathers,
label offsets, the four local_loss x gather_with_grad variants and the reduce-scatter backward - checked
against tests/golden/dist_loss_w{2,8}.npz, which were produced by the REAL reference ClipLoss under 2- and 8-rank
gloo groups.  The HIP kernels cannot run here, so - in this test process only - `clipa_amd.loss.ops` is
replaced by torch-CPU stand-ins built from the oracle maths; the product never takes that route."""
import os
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cpu_ops():
    """Stand-ins with the signatures clipa_amd.loss uses (test infrastructure)."""
    bf16, f32 = torch.bfloat16, torch.float32
    o = types.SimpleNamespace()
    o.to_bf16 = lambda t: t.to(bf16)
    o.transpose_bf16 = lambda t: t.to(bf16).T.contiguous()
    o.gemm_nt = lambda a, b, alpha=1.0, out_f32=True: (a.float() @ b.float().T) * alpha
    o.gemm_tn = lambda p, q, dt=f32: (p.float().T @ q.float()).to(dt)

    def simce(rows, cols, n_valid, label0, gscale, scale=None, want_grad=True):
        R = rows.shape[0]
        n8 = (n_valid + 7) // 8 * 8
        s = float(scale.reshape(-1)[0]) if scale is not None else 1.0
        raw = rows.float() @ cols[:n_valid].float().T
        logits = raw * s
        labels = torch.arange(R) + label0
        loss_rows = torch.logsumexp(logits, dim=1) - logits[torch.arange(R), labels]
        if not want_grad:
            return loss_rows, None, None
        p = torch.softmax(logits, dim=1)
        p[torch.arange(R), labels] -= 1.0
        g = p * gscale
        dl = torch.zeros((R, n8), dtype=bf16)
        dl[:, :n_valid] = (g * s).to(bf16)
        return loss_rows, dl, (g * raw).sum(1)

    def sum_scale(x, scale, out=None, accumulate=False):
        v = x.sum() * scale
        if out is None:
            return v.reshape(())
        out.copy_(out + v if accumulate else v)
        return out

    o.simce, o.sum_scale = simce, sum_scale
    return o


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, ROOT)
    import clipa_amd.loss as L
    L.ops = _cpu_ops()
    z = np.load(os.path.join(ROOT, "tests", "golden", f"dist_loss_w{world}.npz"))
    B = int(z["B"])
    res = {}
    for local_loss in (True, False):
        for gwg in (True, False):
            i = torch.from_numpy(z["img"][rank * B:(rank + 1) * B]).clone().requires_grad_(True)
            t = torch.from_numpy(z["txt"][rank * B:(rank + 1) * B]).clone().requires_grad_(True)
            s = torch.tensor(float(z["logit_scale"]), requires_grad=True)
            fn = L.ClipLoss(local_loss=local_loss, gather_with_grad=gwg, cache_labels=True, rank=rank, world_size=world)
            loss = fn(i, t, s, output_dict=True)["contrastive_loss"]
            loss.backward()
            res[f"{int(local_loss)}{int(gwg)}"] = (float(loss), i.grad.numpy(), t.grad.numpy(), float(s.grad))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,port", [(2, 29741), (8, 29743)])
def test_cliploss_gloo_ranks_match_reference(world, port):
    """2 ranks (B = 8 each) and the node size of BASELINE configs 3-5: 8 ranks with an ODD per-rank batch of 3 (label offset
    3 * rank, 24 gathered rows), every local_loss x gather_with_grad variant, against the REAL reference's ClipLoss under the
    same gloo groups (oracle/make_golden.py run_dist).  On CPU tensors gloo runs the production collectives
    (all_gather_into_tensor on uint8 views, reduce_scatter_tensor of the fused fp32 gradient)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    z = np.load(os.path.join(ROOT, "tests", "golden", f"dist_loss_w{world}.npz"))
    assert int(z["world"]) == world
    for rank in range(world):
        for key, (loss, gi, gt, gs) in got[rank].items():
            # bf16 features / bf16 dlogits in the engine's data path -> bf16-level tolerance
            assert abs(loss - float(z[f"loss_{key}_r{rank}"])) < 2e-2 * abs(float(z[f"loss_{key}_r{rank}"])), (key, rank)
            for a, b in ((gi, z[f"gi_{key}_r{rank}"]), (gt, z[f"gt_{key}_r{rank}"])):
                a, b = a.reshape(-1).astype(np.float64), b.reshape(-1).astype(np.float64)
                cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
                assert cos > 0.999, (key, rank, cos)
                assert abs(np.linalg.norm(a) / np.linalg.norm(b) - 1) < 2e-2, (key, rank)
            assert abs(gs - float(z[f"gs_{key}_r{rank}"])) < 3e-2 * abs(float(z[f"gs_{key}_r{rank}"])) + 1e-4, (key, rank)
