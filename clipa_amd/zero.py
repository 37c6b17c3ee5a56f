"""Sharded gradient exchange + sharded AdamW state for the data-parallel step.

The reference wraps the model in torch DDP (clipa_torch/training/main.py:292-299: bucketed ring all-reduce of every
gradient, AVG) and gives every rank a full torch.optim.AdamW (main.py:318-326: 2 x fp32 moments of ALL parameters on
every GPU, the same update computed W times).  `ShardedAdamW` replaces the pair with the ZeRO-1 decomposition of the same
arithmetic:

  backward   gradients accumulate straight into flat per-bucket buffers; when a bucket is complete (autograd hooks, so the
             exchange overlaps the rest of the backward pass) it is REDUCE-SCATTERED (SUM; the 1/W of DDP's average is
             applied inside the update kernel): rank r receives only the r-th 1/W of the bucket.  `exchange="all_to_all"` does the same exchange as W-1 simultaneous one-hop peer
             sends over the xGMI mesh (all_to_all_single) + a local fp32 sum (clipa_reduce_shards) instead of RCCL's
             reduce-scatter schedule - on MI355X every pair of GPUs has its own link, a ring uses one of seven;
  step       each rank updates ONLY its shard with the fused multi-tensor AdamW kernel (moments exist only for the shard:
             optimizer state / W per GPU), with the global-norm clip computed from the shards (one scalar all-reduce);
  after      the updated parameter shards are ALL-GATHERED into the flat parameter buffers the model's parameters are
             views of, and their version counters are bumped (the engine's bf16 / fp8 weight caches refresh).

Wire bytes per step are those of the ring all-reduce (2 (W-1)/W x gradient bytes) but in two phases that run on every link at
once.  With world_size 1 (or no process group) it degenerates to the plain fused AdamW over flat buffers.  Semantics match
DDP + AdamW: gradients are averaged over ranks in the parameters' dtype, same hyper-parameters per param group
(`param_groups[i]["lr"]` stays assignable by the reference's scheduler, training/scheduler.py), same clamp of logit_scale
(training/train.py:285-286).  `no_sync()` defers the exchange like DDP's (gradient accumulation, train.py:216-256).
"""
import contextlib

import torch
import torch.distributed as dist

from . import ops


class _Bucket:
    __slots__ = ("group", "params", "offsets", "numel", "flat", "grad", "shard", "gshard", "recv", "m", "v", "arrived",
                 "handle", "launched", "stale")


class ShardedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_clip_norm=None, clamp=None,
                 process_group=None, bucket_bytes=256 << 20, exchange="reduce_scatter", broadcast_parameters=True,
                 force_collectives=False, tensor_collectives=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if exchange not in ("reduce_scatter", "all_to_all"):
            raise ValueError(f"exchange={exchange!r}: 'reduce_scatter' or 'all_to_all'")
        self.grad_clip_norm, self.clamp, self.exchange = grad_clip_norm, clamp, exchange
        self.last_grad_norm = None
        self.group = process_group
        self.dist_on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if self.dist_on else 1
        self.rank = dist.get_rank(process_group) if self.dist_on else 0
        # torch's gloo backend (CPU tests, 2 ranks on one GPU) lacks reduce_scatter / *_into_tensor: same maths via all_reduce
        # (tensor_collectives=True walks the production calls - reduce_scatter_tensor / all_to_all_single / all_gather_into_tensor -
        # on a backend that has them for the tensors at hand: gloo with CPU tensors, the 8-rank CPU tests)
        self._tensor_collectives = (self.dist_on and dist.get_backend(process_group) == "nccl") if tensor_collectives is None \
            else bool(tensor_collectives) and self.dist_on
        # force_collectives: issue every collective even in a 1-rank group (they degenerate to copies) - the pre-flight of
        # the RCCL call sequence on a single GPU (tests, bench.py under CLIPA_BENCH_FORCE_DIST=1)
        self._collect = self.world > 1 or (bool(force_collectives) and self.dist_on)
        self._sync = True
        self._steps = 0
        self.buckets = []
        self._build(int(bucket_bytes))
        if broadcast_parameters and self.world > 1:               # what DDP does at construction (rank 0's weights win)
            for b in self.buckets:
                dist.broadcast(b.flat, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                               group=process_group)
            self._bump_versions()

    # ---- layout ---------------------------------------------------------------------------------------------------------
    def _build(self, bucket_bytes):
        align = 8 * self.world                                    # shards stay 16-byte aligned, sizes multiples of 8
        for gi, group in enumerate(self.param_groups):
            by_kind = {}
            for p in group["params"]:
                if p.requires_grad:
                    by_kind.setdefault((p.dtype, p.device), []).append(p)
            for (dtype, device), plist in by_kind.items():
                if dtype not in (torch.float32, torch.bfloat16):
                    raise RuntimeError(f"ShardedAdamW: parameters must be f32 or bf16, got {dtype}")
                esz = 4 if dtype == torch.float32 else 2
                cur, cur_n = [], 0
                for p in reversed(plist):                          # backward produces the last layers' gradients first
                    cur.append(p)
                    cur_n += p.numel()
                    if cur_n * esz >= bucket_bytes:
                        self._make_bucket(gi, cur, dtype, device, align)
                        cur, cur_n = [], 0
                if cur:
                    self._make_bucket(gi, cur, dtype, device, align)

    def _make_bucket(self, gi, plist, dtype, device, align):
        b = _Bucket()
        b.group, b.params, b.offsets = gi, list(plist), []
        off = 0
        for p in plist:
            b.offsets.append(off)
            off += (p.numel() + 7) // 8 * 8                       # every parameter starts on a 16-byte boundary
        b.numel = (off + align - 1) // align * align
        b.flat = torch.zeros(b.numel, dtype=dtype, device=device)
        b.grad = torch.zeros(b.numel, dtype=dtype, device=device)
        with torch.no_grad():
            for p, o in zip(plist, b.offsets):
                view = b.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view                                      # the model's parameter now lives in the flat buffer
                p.grad = b.grad[o:o + p.numel()].view(p.shape)     # autograd accumulates in place into the flat gradient
        n = b.numel // self.world
        lo = self.rank * n
        b.shard = b.flat[lo:lo + n]
        b.gshard = torch.zeros(n, dtype=dtype, device=device) if self._collect else b.grad[lo:lo + n]
        b.recv = None
        b.m = torch.zeros(n, dtype=torch.float32, device=device)
        b.v = torch.zeros(n, dtype=torch.float32, device=device)
        b.arrived, b.handle, b.launched, b.stale = 0, None, False, False
        for p in plist:
            p.register_post_accumulate_grad_hook(lambda _p, b=b: self._on_grad(b))
        self.buckets.append(b)

    # ---- gradient exchange ----------------------------------------------------------------------------------------------
    def _on_grad(self, b):
        """A gradient of bucket `b` has been accumulated.  The reference's accum_freq loop (train.py:243-256) calls backward()
        several times per optimizer step with no no_sync(): a backward that lands after the bucket's exchange was launched
        makes that exchange stale, and every completed round of arrivals (re-)exchanges the CUMULATIVE flat gradient - what
        DDP does when it all-reduces on every backward.  Inside no_sync() nothing is exchanged; step() catches up."""
        b.arrived += 1
        if b.launched:
            b.stale = True
        if self._sync and b.arrived % len(b.params) == 0:
            self._launch(b)

    def _launch(self, b):
        # Invariant: an exchange that a later backward of the same step makes stale is never CONSUMED - its handle is
        # waited for here (the collective may still be reading b.grad while that backward accumulates into it; the result is
        # discarded) and step() only reads gshard / recv of the LAST launch (`_finish` relaunches a stale bucket first).  The
        # plain accum_freq loop therefore moves accum_freq x the bytes; `no_sync()` around all but the last backward avoids it.
        if b.handle is not None:                                  # superseded exchange of an earlier backward: let it drain
            b.handle.wait()
            b.handle = None
        b.launched, b.stale = True, False
        if not self._collect:
            return
        if not self._tensor_collectives:
            # gloo (tests): all-reduce a COPY - in place it would fold other ranks' sums into the local accumulator that a
            # later backward of the same step keeps adding to
            b.recv = b.grad.clone()
            b.handle = dist.all_reduce(b.recv, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        elif self.exchange == "all_to_all":
            if b.recv is None:
                b.recv = torch.empty_like(b.grad)
            b.handle = dist.all_to_all_single(b.recv, b.grad, group=self.group, async_op=True)
        else:
            # SUM on the wire, the 1/W of DDP's average rides in the update kernel's grad_scale (as DDP pre-divides): RCCL's AVG
            # path dropped the tail of an fp32 bucket on a 1-rank group (tools/zero_diag.py, profiles/r02_zero_diag.txt)
            b.handle = dist.reduce_scatter_tensor(b.gshard, b.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _finish(self, b):
        if not b.launched or b.stale:                             # no gradient hook completed a round (unused parameters,
            self._launch(b)                                       # no_sync), or gradients arrived after the last exchange
        if b.handle is not None:
            b.handle.wait()
            b.handle = None
            n = b.numel // self.world
            if not self._tensor_collectives:
                b.gshard.copy_(b.recv[self.rank * n:(self.rank + 1) * n])
                b.recv = None
            elif self.exchange == "all_to_all":
                ops.reduce_shards(b.recv, self.world, out=b.gshard, scale=1.0)

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside accumulate into the flat gradients without exchanging them (DDP.no_sync)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def zero_grad(self, set_to_none=False):
        """Gradients are views of the flat buffers: one memset per bucket, never None."""
        for b in self.buckets:
            if b.handle is not None:                              # an exchange nobody consumed (zero_grad without step)
                b.handle.wait()
                b.handle = None
            b.grad.zero_()
            b.arrived, b.launched, b.stale = 0, False, False
            for p, o in zip(b.params, b.offsets):
                if p.grad is None or p.grad.data_ptr() != b.grad.data_ptr() + o * b.grad.element_size():
                    p.grad = b.grad[o:o + p.numel()].view(p.shape)

    # ---- update -----------------------------------------------------------------------------------------------------------
    def _bump_versions(self):
        for b in self.buckets:
            for p in b.params:
                torch.autograd.graph.increment_version(p)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for b in self.buckets:
            self._finish(b)
            assert not b.stale and b.handle is None, "ShardedAdamW: a stale gradient exchange reached the update"
        # the shards hold the SUM over ranks; DDP's average = sum / W is applied as grad_scale inside the update kernel
        avg = 1.0 / self.world if self._collect else 1.0
        coef = None
        if self.grad_clip_norm is not None and self.buckets:
            buf = ops.grad_sqnorm([b.gshard for b in self.buckets])   # padding is zero: shards partition the gradient
            if self._collect:
                dist.all_reduce(buf[0:1], op=dist.ReduceOp.SUM, group=self.group)
            if avg != 1.0:
                buf[0:1].mul_(avg * avg)                              # |g / W|^2
            self.last_grad_norm, coef = ops.clip_coef(buf, self.grad_clip_norm)
        self._steps += 1
        calls = {}
        for b in self.buckets:
            calls.setdefault((b.group, b.flat.dtype, b.flat.device), []).append(b)
        for (gi, _, _), bs in calls.items():
            g = self.param_groups[gi]
            ops.adamw_multi_([b.shard for b in bs], [b.gshard for b in bs], [b.m for b in bs], [b.v for b in bs],
                             lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"],
                             weight_decay=g["weight_decay"], step=self._steps, grad_scale=avg, grad_scale_dev=coef)
        if self._collect:
            handles = []
            for b in self.buckets:
                if self._tensor_collectives:
                    handles.append(dist.all_gather_into_tensor(b.flat, b.shard, group=self.group, async_op=True))
                else:
                    n = b.numel // self.world
                    outs = [b.flat[r * n:(r + 1) * n] for r in range(self.world)]
                    handles.append(dist.all_gather(outs, b.shard.clone(), group=self.group, async_op=True))
            for h in handles:
                h.wait()
        if self.clamp is not None:
            self.clamp[0].data.clamp_(self.clamp[1], self.clamp[2])
        self._bump_versions()
        for b in self.buckets:
            b.arrived, b.launched, b.stale = 0, False, False
        return loss

    # ---- checkpoint interchange (training/main.py:338-356, 436-468): the layout of torch.optim.AdamW.state_dict() ----------
    def _gather_moment(self, b, t):
        if not self._collect:
            return t
        full = torch.empty(b.numel, dtype=torch.float32, device=t.device)
        if self._tensor_collectives:
            dist.all_gather_into_tensor(full, t, group=self.group)
        else:
            n = b.numel // self.world
            dist.all_gather([full[r * n:(r + 1) * n] for r in range(self.world)], t, group=self.group)
        return full

    def state_dict(self):
        """COLLECTIVE: every rank must call it (the moments are sharded; this all-gathers them) - full, un-sharded state in
        torch.optim.AdamW's format.  The reference builds its checkpoint dict only on the master rank (main.py:437-443,
        inside `if args.save_logs`): with this optimizer hoist `opt_state = optimizer.state_dict()` above that branch
        (INTEGRATION.md section 1), or rank 0 blocks in the gather while the other ranks train on."""
        index = {}
        k = 0
        for g in self.param_groups:
            for p in g["params"]:
                index[id(p)] = k
                k += 1
        state = {}
        for b in self.buckets:
            m, v = self._gather_moment(b, b.m), self._gather_moment(b, b.v)
            for p, o in zip(b.params, b.offsets):
                state[index[id(p)]] = {"step": torch.tensor(float(self._steps)),
                                       "exp_avg": m[o:o + p.numel()].view(p.shape).clone(),
                                       "exp_avg_sq": v[o:o + p.numel()].view(p.shape).clone()}
        groups = []
        k = 0
        for g in self.param_groups:
            d = {key: val for key, val in g.items() if key != "params"}
            d["params"] = list(range(k, k + len(g["params"])))
            k += len(g["params"])
            groups.append(d)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        index = {}
        k = 0
        for g, sg in zip(self.param_groups, state_dict["param_groups"]):
            for key, val in sg.items():
                if key != "params":
                    g[key] = val
            for p in g["params"]:
                index[id(p)] = k
                k += 1
        steps = 0
        for b in self.buckets:
            n = b.numel // self.world
            lo = self.rank * n
            m = torch.zeros(b.numel, dtype=torch.float32, device=b.m.device)
            v = torch.zeros(b.numel, dtype=torch.float32, device=b.m.device)
            for p, o in zip(b.params, b.offsets):
                st = state_dict["state"].get(index[id(p)])
                if st is None:
                    continue
                m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1).to(torch.float32))
                v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1).to(torch.float32))
                steps = max(steps, int(float(st["step"])))
            b.m.copy_(m[lo:lo + n])
            b.v.copy_(v[lo:lo + n])
        self._steps = steps
