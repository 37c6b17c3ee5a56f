"""fp8 vs bf16 GEMM on the production launch shapes (in-process interleaved A/B, HIP-event timed) + the quantisers.
    python tools/f8_bench.py [--quick] > gpurun_out/f8_bench.jsonl
Shapes: ViT-L/16 @ 224 local batch 4096 (M = 806 912) and ViT-H/14 @ 224 local batch 2048 (M = 526 336)."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops

bf16 = torch.bfloat16
DEV = "cuda"


def emit(**kw):
    print(json.dumps(kw), flush=True)


def timed(fn, iters):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def run(M, N, K, epi, rounds, iters):
    torch.manual_seed(1)
    a = torch.randn(M, K, device=DEV).to(bf16)
    w = (torch.randn(N, K, device=DEV) * 0.05).to(bf16)
    bias = torch.randn(N, device=DEV)
    aux = torch.randn(M, N, device=DEV).to(bf16) if epi in ("res", "dact") else None
    qa, sa = ops.quantize_rows(a)
    qw, sw = ops.quantize_rows(w)
    kw = {"bias": {}, "gelu": dict(epi=ops.EPI_ACT), "gelu+pre": dict(epi=ops.EPI_ACT, want_pre=True),
          "res": dict(epi=ops.EPI_ADD, aux=aux), "dact": dict(epi=ops.EPI_DACT, aux=aux), "none": {}}[epi]
    b_ = None if epi in ("none", "dact") else bias
    f_bf = lambda: ops.gemm_nt(a, w, b_, **kw)
    f_f8 = lambda: ops.gemm_nt_f8(qa, sa, qw, sw, b_, **kw)
    f_q = lambda: ops.quantize_rows(a)
    for f in (f_bf, f_f8, f_q):
        f(); f()
    res = {"bf16": [], "fp8": [], "quant": []}
    for _ in range(rounds):
        for name, f in (("bf16", f_bf), ("fp8", f_f8), ("quant", f_q)):
            res[name].append(timed(f, iters)[0])
    fl = 2.0 * M * N * K
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    emit(kind="gemm", M=M, N=N, K=K, epi=epi, bf16_ms=round(med["bf16"], 3), fp8_ms=round(med["fp8"], 3),
         bf16_tf=round(fl / med["bf16"] / 1e9, 1), fp8_tf=round(fl / med["fp8"] / 1e9, 1),
         quant_a_ms=round(med["quant"], 3), quant_a_tbs=round(3.0 * M * K / med["quant"] / 1e9, 2),
         speedup=round(med["bf16"] / med["fp8"], 3), speedup_incl_quant=round(med["bf16"] / (med["fp8"] + med["quant"]), 3))
    del a, w, aux, qa, qw
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    rounds, iters = (2, 3) if args.quick else (3, 5)
    ML, MH = 806912, 526336
    shapes = [(ML, 4096, 1024, "gelu"), (ML, 4096, 1024, "gelu+pre"), (ML, 4096, 1024, "dact"), (ML, 1024, 4096, "res"),
              (ML, 1024, 4096, "none"), (ML, 3072, 1024, "bias"), (ML, 1024, 3072, "none"), (ML, 1024, 1024, "res"),
              (MH, 5120, 1280, "gelu"), (MH, 1280, 5120, "res"), (MH, 3840, 1280, "bias"), (MH, 1280, 1280, "res")]
    if args.quick:
        shapes = shapes[:1] + shapes[3:4] + shapes[8:9]
    for s in shapes:
        run(*s, rounds, iters)
    # main loop only (ablation flag 2: no epilogue) on the headline shape
    import ctypes
    from clipa_amd import lib
    h = lib.load()
    lib.debug_set(0, 2)
    run(ML, 4096, 1024, "none", rounds, iters)
    lib.debug_set(0, 0)
    # LayerNorm with / without the fused fp8 output
    x = torch.randn(ML, 1024, device=DEV).to(bf16)
    g, b = torch.ones(1024, device=DEV), torch.zeros(1024, device=DEV)
    t_ln = timed(lambda: ops.layernorm_fwd(x, g, b), 5)[0]
    t_q8 = timed(lambda: ops.layernorm_fwd_q8(x, g, b), 5)[0]
    t_q8y = timed(lambda: ops.layernorm_fwd_q8(x, g, b, want_bf16=True), 5)[0]
    emit(kind="layernorm", rows=ML, D=1024, ln_ms=round(t_ln, 3), ln_q8_ms=round(t_q8, 3), ln_q8_bf16_ms=round(t_q8y, 3))


if __name__ == "__main__":
    main()
