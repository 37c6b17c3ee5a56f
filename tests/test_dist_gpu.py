"""Two ranks, ONE GPU: the whole data-parallel training step (engine autograd nodes under the reference's DDP
wrapper + ClipLoss local_loss/gather_with_grad with its fused all-gather / reduce-scatter backward) against
the single-process global-batch step.  The test boxes have one GPU, where RCCL refuses two ranks per device,
so the transport here is gloo on device tensors; everything around the transport is the product path."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

B_LOC, S, CTX = 8, 112, 32


def _build(dev, precision="amp_bf16"):
    import clipa_amd
    torch.manual_seed(0)
    m = clipa_amd.create_model("ViT-S-16", precision=precision, device=dev, force_image_size=S, output_dict=True)
    m.positional_embedding = torch.nn.Parameter(m.positional_embedding[:CTX].clone())
    m.set_grad_checkpointing(True)
    m.transformer.keep_blocks = 2          # mix recomputed and kept blocks
    return m


def _batch(world):
    from oracle import clip_oracle as O
    img, txt = O.synthetic_batch(B_LOC * world, S, CTX, 49408, seed=7)
    return img, txt


def _get(q, procs, limit=300):
    """Queue read that gives up as soon as a worker has died (a crashed rank must not stall the suite)."""
    import queue
    import time
    t0 = time.time()
    while True:
        try:
            return q.get(timeout=2)
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs) or time.time() - t0 > limit:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise AssertionError("a worker rank died or timed out: " + str([p.exitcode for p in procs]))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, ROOT)
    import clipa_amd
    model = _build(dev)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], static_graph=True)
    loss_fn = clipa_amd.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True, rank=rank, world_size=world)
    img, txt = _batch(world)
    img = img[rank * B_LOC:(rank + 1) * B_LOC].to(dev)
    txt = txt[rank * B_LOC:(rank + 1) * B_LOC].to(dev)
    losses = []
    for _ in range(2):                      # second step exercises static_graph's cached bucket order
        ddp.zero_grad(set_to_none=True)
        out = ddp(img, txt)
        loss = loss_fn(**out, output_dict=True)["contrastive_loss"]
        loss.backward()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().float().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None}
    q.put((rank, losses, grads))          # numpy: pickled by value (torch tensors travel as fds that die with the worker)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_global_batch_step():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, 29763, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, losses, grads = _get(q, procs)
        got[rank] = (losses, grads)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0

    import clipa_amd
    dev = torch.device("cuda", 0)
    model = _build(dev)
    img, txt = _batch(world)
    out = model(img.to(dev), txt.to(dev))
    loss = clipa_amd.ClipLoss()(**out, output_dict=True)["contrastive_loss"]
    loss.backward()
    ref = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}

    # global loss = mean of the ranks' local-loss values; DDP's averaged gradient = gradient of the global loss
    mean_local = 0.5 * (got[0][0][0] + got[1][0][0])
    assert abs(mean_local - float(loss)) <= 3e-3 * abs(float(loss)), (mean_local, float(loss))
    assert abs(got[0][0][0] - got[0][0][1]) < 1e-6          # same weights, same batch -> same loss on step 2
    assert set(got[0][1]) == set(ref)
    for n, g in ref.items():
        a, b = torch.from_numpy(got[0][1][n]), torch.from_numpy(got[1][1][n])
        assert torch.equal(a, b), f"ranks disagree on {n} after the all-reduce"
        cos = torch.nn.functional.cosine_similarity(a.flatten(), g.flatten(), dim=0).item()
        rel = (a.norm() / g.norm().clamp_min(1e-12)).item()
        assert cos >= 0.99 and 0.95 <= rel <= 1.05, (n, cos, rel)


def _rccl_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import sys
    sys.path.insert(0, ROOT)
    import clipa_amd
    import clipa_amd.loss as L
    local = torch.randn(64, 256, device=dev).to(torch.bfloat16)
    out = L._gather_fused(local, 1)
    full = torch.randn(64, 256, device=dev)
    rs = L._reduce_scatter_fused(full, 1)
    torch.cuda.synchronize()
    ok = torch.equal(out, local) and torch.equal(rs, full)
    # the whole step under DDP on the RCCL backend (world 1: bucket all-reduce is the identity), pure-bf16
    # parameters and gradients as in bench.py
    model = _build(dev, precision="bf16")
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], static_graph=True)
    img, txt = _batch(1)
    o = ddp(img.to(dev), txt.to(dev))
    loss = clipa_amd.ClipLoss(local_loss=True, gather_with_grad=True, rank=0, world_size=1)(**o, output_dict=True)["contrastive_loss"]
    loss.backward()
    torch.cuda.synchronize()
    ok = ok and bool(torch.isfinite(loss)) and all(p.grad is not None for p in model.parameters() if p.requires_grad)
    q.put(ok)
    dist.destroy_process_group()


def test_rccl_backend_single_rank_collectives_and_ddp():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29771, q))
    p.start()
    assert _get(q, [p]) is True
    p.join(timeout=120)
    assert p.exitcode == 0
