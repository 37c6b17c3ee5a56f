"""Staggered-epilogue GEMM (clipa_internal_debug_set variant 21 / 22) vs the production kernel: parity against fp64 on small forced
shapes, every epilogue, then an interleaved A/B on the production launch shapes.   python tools/nts_check.py [--quick]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops, lib
from oracle import clip_oracle as O
h = lib.load()
bf16, f32 = torch.bfloat16, torch.float32
DEV = "cuda"

def emit(**kw): print(json.dumps(kw), flush=True)

def worst(got, ref, rtol, atol):
    got, ref = got.double().cpu(), ref.double()
    err = (got - ref).abs() - (atol + rtol * ref.abs())
    return float(err.max()), int((err > 0).sum())

def parity():
    bad = 0
    for (M, N, K) in [(2000, 520, 776), (4096, 1024, 1024), (9000, 1280, 640), (777, 264, 3072), (300, 256, 576), (70000, 768, 768)]:
        g = torch.Generator().manual_seed(M + N)
        a = torch.randn(M, K, generator=g).to(bf16); b = (torch.randn(N, K, generator=g) * 0.05).to(bf16)
        bias = torch.randn(N, generator=g); aux = torch.randn(M, N, generator=g).to(bf16)
        A, B, BIAS, AUX = a.to(DEV), b.to(DEV), bias.to(DEV), aux.to(DEV)
        rows = torch.randperm(M, generator=g)[:1500] if M > 3000 else torch.arange(M)
        lin = a[rows].double() @ b.double().T + bias.double()
        prev = None
        for rep in range(2):
            lib.debug_set(22, 0)
            o_bias = ops.gemm_nt(A, B, BIAS)
            o_add = ops.gemm_nt(A, B, BIAS, epi=ops.EPI_ADD, aux=AUX)
            o_act, o_pre = ops.gemm_nt(A, B, BIAS, epi=ops.EPI_ACT, act=0, want_pre=True)
            o_act1 = ops.gemm_nt(A, B, BIAS, epi=ops.EPI_ACT, act=0)
            o_dact = ops.gemm_nt(A, B, BIAS, epi=ops.EPI_DACT, act=0, aux=AUX)
            lib.debug_set(0, 0)
            torch.cuda.synchronize()
            res = {}
            res["bias"] = worst(o_bias[rows.to(DEV)], lin, 2 ** -7, 2e-3)
            res["add"] = worst(o_add[rows.to(DEV)], lin + aux[rows].double(), 2 ** -6, 2e-2)
            res["pre"] = worst(o_pre[rows.to(DEV)], lin, 2 ** -7, 2e-3)
            res["act"] = worst(o_act[rows.to(DEV)], O.activation(o_pre[rows.to(DEV)].double().cpu(), "gelu_erf"), 2 ** -7, 2e-3)
            x = aux[rows].double().clone().requires_grad_(True); O.activation(x, "gelu_erf").sum().backward()
            res["dact"] = worst(o_dact[rows.to(DEV)], lin.to(bf16).double() * x.grad, 2 ** -6, 6e-3)
            same12 = bool(torch.equal(o_act, o_act1))
            got = (o_bias, o_add, o_act, o_pre, o_dact)
            rerun = True if prev is None else all(torch.equal(x_, y_) for x_, y_ in zip(prev, got))
            prev = got
            nb = sum(v[1] for v in res.values()) + (not same12) + (not rerun)
            bad += nb
            emit(kind="parity", M=M, N=N, K=K, rep=rep, bad=nb, act_1out_equals_2out=same12, relaunch_equal=rerun,
                 **{k: round(v[0], 5) for k, v in res.items()})
    return bad

def timed(fn, iters=5):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]

def bench(quick):
    M = 806912
    shapes = [(4096, 1024, "gelu"), (4096, 1024, "gelu+pre"), (4096, 1024, "dact"), (1024, 4096, "res"), (3072, 1024, "bias"), (1024, 1024, "res"), (1024, 3072, "none")]
    if quick: shapes = shapes[:2] + shapes[3:4]
    for (N, K, epi) in shapes:
        torch.manual_seed(1)
        a = torch.randn(M, K, device=DEV).to(bf16); w = (torch.randn(N, K, device=DEV) * 0.05).to(bf16)
        bias = torch.randn(N, device=DEV); aux = torch.randn(M, N, device=DEV).to(bf16) if epi in ("res", "dact") else None
        kw = {"bias": {}, "gelu": dict(epi=ops.EPI_ACT), "gelu+pre": dict(epi=ops.EPI_ACT, want_pre=True), "res": dict(epi=ops.EPI_ADD, aux=aux),
              "dact": dict(epi=ops.EPI_DACT, aux=aux), "none": {}}[epi]
        b_ = None if epi in ("none", "dact") else bias
        f = lambda: ops.gemm_nt(a, w, b_, **kw)
        t = {0: [], 21: []}
        outs = {}
        for rnd in range(3):
            for var in (0, 21):
                lib.debug_set(var, 0)
                outs[var] = f(); f()
                t[var].append(timed(f))
        lib.debug_set(0, 0)
        o0 = outs[0][0] if isinstance(outs[0], tuple) else outs[0]
        o1 = outs[21][0] if isinstance(outs[21], tuple) else outs[21]
        diff = float((o0.float() - o1.float()).abs().max())
        fl = 2.0 * M * N * K
        t0, t1 = sorted(t[0])[1], sorted(t[21])[1]
        emit(kind="ab", M=M, N=N, K=K, epi=epi, prod_ms=round(t0, 3), nts_ms=round(t1, 3), prod_tf=round(fl / t0 / 1e9, 1),
             nts_tf=round(fl / t1 / 1e9, 1), speedup=round(t0 / t1, 3), max_abs_diff=diff)
        del a, w, aux, outs
        torch.cuda.empty_cache()

if __name__ == "__main__":
    quick = "--quick" in sys.argv
    bad = parity()
    emit(kind="parity_total", bad=bad)
    bench(quick)
