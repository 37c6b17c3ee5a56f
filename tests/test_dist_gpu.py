"""Two ranks, ONE GPU: the whole data-parallel training step (engine autograd nodes under the reference's DDP
wrapper + ClipLoss with its early image-feature all-gather / reduce-scatter backward) against the single-process
global-batch step; the four local_loss x gather_with_grad variants of the HIP ClipLoss against the golden vectors of
the REAL reference under a 2-rank group; the accum_freq = 2 feature-cache step of train.py:216-256.  The test boxes have
one GPU, where RCCL refuses two ranks per device, so the transport here is gloo on device tensors; everything around
the transport is the product path (libclipa_hip.so kernels)."""
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


def _accum_step(ddp, loss_fn, img, txt, accum, lens=None):
    """training/train.py:216-256 restated: no-grad forward of every micro-batch caching the features, then per
    micro-batch a forward WITH grad whose features are spliced into the cached list, the full loss, backward."""
    feats = {"image_features": [], "text_features": []}
    lens = [c.tolist() for c in torch.as_tensor(lens).chunk(accum)] if lens is not None else [None] * accum      # lists: DDP leaves them on the host
    chunks = list(zip(img.chunk(accum), txt.chunk(accum), lens))
    with torch.no_grad():
        for im, tx, ln in chunks:
            out = ddp(im, tx, text_lengths=ln)
            for k in feats:
                feats[k].append(out[k])
    losses = []
    for j, (im, tx, ln) in enumerate(chunks):
        out = ddp(im, tx, text_lengths=ln)
        scale = out.pop("logit_scale")
        inputs = {k: torch.cat(v[:j] + [out[k]] + v[j + 1:]) for k, v in feats.items()}
        loss = loss_fn(**inputs, logit_scale=scale, output_dict=True)["contrastive_loss"]
        loss.backward()
        losses.append(float(loss.detach()))
    return losses


def _worker(rank, world, port, q, accum=1, unpad=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, ROOT)
    import clipa_amd
    model = _build(dev)
    model.unpad_text = bool(unpad)          # decided BEFORE the wrap: DDP(static_graph=True) records the graph of its first iteration
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], static_graph=True)
    loss_fn = clipa_amd.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True, rank=rank, world_size=world).bind(ddp)
    img, txt = _batch(world)
    img = img[rank * B_LOC:(rank + 1) * B_LOC].to(dev)
    txt = txt[rank * B_LOC:(rank + 1) * B_LOC]
    # the caption lengths come from the host side of the loader (no device read-back): the ranks' packed text matrices have
    # different row counts - the towers are rank-local, no collective sees them
    lens = (txt.argmax(-1) + 1).cpu().tolist() if unpad else None      # host-side list (a tensor kwarg would be moved to the GPU by DDP)
    txt = txt.to(dev)
    losses = []
    for _ in range(2):                      # second step exercises static_graph's cached bucket order
        ddp.zero_grad(set_to_none=True)
        if accum > 1:
            losses.append(_accum_step(ddp, loss_fn, img, txt, accum, lens)[-1])
            continue
        out = ddp(img, txt, text_lengths=lens)
        loss = loss_fn(**out, output_dict=True)["contrastive_loss"]
        loss.backward()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().float().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None}
    q.put((rank, losses, grads))          # numpy: pickled by value (torch tensors travel as fds that die with the worker)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("accum,unpad", [(1, False), (2, False), (1, True), (2, True)])
def test_two_rank_step_equals_global_batch_step(accum, unpad):
    """accum = 1: the plain DDP step.  accum = 2: the grad-accumulation "feature cache" step - two
    micro-batches per rank, each re-forwarded with grad against the cached features of the other, two backward passes
    through DDP(static_graph=True); its accumulated gradient is the gradient of the one global-batch loss.
    unpad: the same with the engine's `unpad_text` knob on every rank (set before the DDP wrap, caption lengths from the host):
    the packed text matrices differ in shape between ranks and between micro-batches, the step still equals the padded
    single-process global-batch step."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, 29763 + accum + 4 * int(unpad), q, accum, unpad)) for r in range(world)]
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
        # reference quirk reproduced: every micro-batch backward of train.py:246-256 differentiates the FULL loss
        # w.r.t. logit_scale, so its gradient accumulates accum_freq times
        want = float(accum) if n == "logit_scale" else 1.0
        assert cos >= 0.99 and 0.95 * want <= rel <= 1.05 * want, (n, cos, rel)


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
    out, ev = L._all_gather_bf16(local, 1)
    torch.cuda.current_stream().wait_event(ev)
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


def _sharded_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, ROOT)
    import clipa_amd
    from clipa_amd.zero import ShardedAdamW
    model = _build(dev)
    opt = ShardedAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.95), eps=1e-6, weight_decay=0.1,
                       grad_clip_norm=1.0, bucket_bytes=8 << 20)
    loss_fn = clipa_amd.ClipLoss(local_loss=True, gather_with_grad=True, cache_labels=True, rank=rank, world_size=world).bind(model)
    img, txt = _batch(world)
    img = img[rank * B_LOC:(rank + 1) * B_LOC].to(dev)
    txt = txt[rank * B_LOC:(rank + 1) * B_LOC].to(dev)
    losses = []
    for _ in range(2):
        opt.zero_grad()
        out = model(img, txt)                 # no DDP wrapper: the optimizer's gradient hooks run the exchange
        loss = loss_fn(**out, output_dict=True)["contrastive_loss"]
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    q.put((rank, losses, {n: p.detach().float().cpu().numpy() for n, p in model.named_parameters()}, float(opt.last_grad_norm)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_optimizer_equals_global_batch_adamw():
    """clipa_amd.zero.ShardedAdamW (gradient reduce-scatter + sharded AdamW state + parameter all-gather in
    place of DDP + a full optimizer per rank) on the HIP kernels: two ranks end with identical weights that follow the
    single-process global-batch AdamW run."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, 29781, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, losses, params, norm = _get(q, procs)
        got[rank] = (losses, params, norm)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    import clipa_amd
    from clipa_amd.optim import AdamW
    dev = torch.device("cuda", 0)
    model = _build(dev)
    opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.95), eps=1e-6, weight_decay=0.1,
                grad_clip_norm=1.0)
    img, txt = _batch(world)
    ref_losses = []
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        out = model(img.to(dev), txt.to(dev))
        loss = clipa_amd.ClipLoss()(**out, output_dict=True)["contrastive_loss"]
        loss.backward()
        opt.step()
        ref_losses.append(float(loss.detach()))
    for n, p in model.named_parameters():
        a, b = got[0][1][n], got[1][1][n]
        assert (a == b).all(), f"ranks disagree on {n} after the parameter all-gather"
        # AdamW's normalised update moves a weight by at most ~lr per step whatever the gradient's size
        assert abs(torch.from_numpy(a) - p.detach().float().cpu()).max() <= 2.5e-3, n
    for k in range(2):
        mean_local = 0.5 * (got[0][0][k] + got[1][0][k])
        assert abs(mean_local - ref_losses[k]) <= 5e-3 * abs(ref_losses[k]), (k, mean_local, ref_losses[k])
    assert abs(got[0][2] - got[1][2]) < 1e-5 and abs(got[0][2] - float(opt.last_grad_norm)) < 0.03 * float(opt.last_grad_norm)


def _rccl_sharded_worker(port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import sys
    sys.path.insert(0, ROOT)
    import clipa_amd
    from clipa_amd.optim import AdamW
    from clipa_amd.zero import ShardedAdamW
    img, txt = _batch(1)
    img, txt = img.to(dev), txt.to(dev)
    res, norms = {}, {}
    # without clipping the update is elementwise: bit-equal to the plain optimizer after two steps.  With clipping the norm
    # is summed in a different order (shards vs tensors), so the coefficient may differ in its last bits: ONE step, the norms
    # agree to 1e-5 and every weight to one bf16 ulp (a second step would amplify the ulp through AdamW's sign-like update)
    for clip, steps in ((None, 2), (1.0, 1)):
        for mode in ("plain", "reduce_scatter", "all_to_all"):
            model = _build(dev, precision="bf16")            # pure-bf16 parameters and gradients, as bench.py trains
            params = [p for p in model.parameters() if p.requires_grad]
            kw = dict(lr=1e-3, betas=(0.9, 0.95), eps=1e-6, weight_decay=0.1, grad_clip_norm=clip)
            opt = AdamW(params, **kw) if mode == "plain" else ShardedAdamW(params, bucket_bytes=8 << 20, exchange=mode,
                                                                           force_collectives=True, **kw)
            for _ in range(steps):
                opt.zero_grad()
                out = model(img, txt)
                loss = clipa_amd.ClipLoss(local_loss=True, gather_with_grad=True, rank=0, world_size=1)(**out, output_dict=True)["contrastive_loss"]
                loss.backward()
                opt.step()
            torch.cuda.synchronize()
            res[(clip, mode)] = {n: p.detach().float().cpu() for n, p in model.named_parameters()}
            norms[(clip, mode)] = float(opt.last_grad_norm) if clip else 0.0
    bad = []
    for m in ("reduce_scatter", "all_to_all"):
        for n, ref in res[(None, "plain")].items():
            if not torch.equal(ref, res[(None, m)][n]):
                bad.append((m, "no clip", n, float((ref - res[(None, m)][n]).abs().max())))
        if abs(norms[(1.0, m)] - norms[(1.0, "plain")]) > 1e-5 * norms[(1.0, "plain")]:
            bad.append((m, "clip norm", norms[(1.0, m)], norms[(1.0, "plain")]))
        for n, ref in res[(1.0, "plain")].items():
            d = (ref - res[(1.0, m)][n]).abs()
            if not bool((d <= ref.abs() * 2.0 ** -7 + 1e-6).all()):
                bad.append((m, "clip", n, float(d.max())))
    q.put(bad[:8] if bad else True)
    dist.destroy_process_group()


def test_rccl_single_rank_sharded_optimizer_collectives():
    """The RCCL call sequence of ShardedAdamW - reduce_scatter_tensor(AVG) on bf16 buckets, all_to_all_single + the
    clipa_reduce_shards kernel, the scalar all-reduce of the clip norm, the in-place all_gather_into_tensor - on a 1-rank
    group, where every collective is a copy: the result must equal the plain fused AdamW bit for bit."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_sharded_worker, args=(29783, q))
    p.start()
    got = _get(q, [p])
    assert got is True, got
    p.join(timeout=120)
    assert p.exitcode == 0


def _variant_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, ROOT)
    import numpy as np
    import clipa_amd
    z = np.load(os.path.join(ROOT, "tests", "golden", "dist_loss_w2.npz"))
    B = int(z["B"])
    res = {}
    for local_loss in (True, False):
        for gwg in (True, False):
            i = torch.from_numpy(z["img"][rank * B:(rank + 1) * B]).to(dev).requires_grad_(True)
            t = torch.from_numpy(z["txt"][rank * B:(rank + 1) * B]).to(dev).requires_grad_(True)
            s = torch.tensor(float(z["logit_scale"]), device=dev, requires_grad=True)
            fn = clipa_amd.ClipLoss(local_loss=local_loss, gather_with_grad=gwg, cache_labels=True, rank=rank, world_size=world)
            loss = fn(i, t, s, output_dict=True)["contrastive_loss"]
            loss.backward()
            torch.cuda.synchronize()
            res[f"{int(local_loss)}{int(gwg)}"] = (float(loss), i.grad.cpu().numpy(), t.grad.cpu().numpy(), float(s.grad))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_cliploss_variants_two_ranks_hip_kernels_match_reference_golden():
    """All four local_loss x gather_with_grad variants of the HIP ClipLoss (similarity GEMMs, cross-entropy kernel,
    gradient GEMMs, gather / reduce-scatter glue) under a 2-rank group against tests/golden/dist_loss_w2.npz, which the
    REAL reference ClipLoss produced under a 2-rank gloo group: loss, d/d image features, d/d text features, d/d scale."""
    import numpy as np
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_variant_worker, args=(r, world, 29781, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(_get(q, procs) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    z = np.load(os.path.join(ROOT, "tests", "golden", "dist_loss_w2.npz"))
    for rank in range(world):
        for key, (loss, gi, gt, gs) in got[rank].items():
            ref_loss = float(z[f"loss_{key}_r{rank}"])
            assert abs(loss - ref_loss) < 2e-2 * abs(ref_loss), (key, rank, loss, ref_loss)   # bf16 features on the wire
            for a, b in ((gi, z[f"gi_{key}_r{rank}"]), (gt, z[f"gt_{key}_r{rank}"])):
                a, b = a.reshape(-1).astype(np.float64), b.reshape(-1).astype(np.float64)
                cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
                assert cos > 0.999, (key, rank, cos)
                assert abs(np.linalg.norm(a) / np.linalg.norm(b) - 1) < 2e-2, (key, rank)
            ref_gs = float(z[f"gs_{key}_r{rank}"])
            assert abs(gs - ref_gs) < 3e-2 * abs(ref_gs) + 1e-4, (key, rank, gs, ref_gs)
