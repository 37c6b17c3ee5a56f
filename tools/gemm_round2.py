"""Round-2 gemm_nt experiment: race screen + interleaved in-process A/B of the bf16-output kernels on the production
launch shapes (ViT-L/16 @ 224, local batch 4096: M = 806 912; text tower M = 315 392).
    python tools/gemm_round2.py [--quick] > gpurun_out/gemm_round2.jsonl
Variants are "nt:abl" pairs for clipa_internal_debug_set: nt 11 LDS-window epilogue (round 1), 12 direct epilogue (13 / 14 /
15 = the experiments of tools/experiments/gemm_nt_round2.hip when pasted back); abl 1 = epilogue maths without
stores, 2 = main loop only."""
import argparse, ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops, lib

h = lib.load()
bf16 = torch.bfloat16
DEV = "cuda"


def emit(**kw):
    print(json.dumps(kw), flush=True)


def call(epi, a, w, bias, aux):
    if epi == "gelu+pre": return ops.gemm_nt(a, w, bias, epi=ops.EPI_ACT, want_pre=True)
    if epi == "gelu": return ops.gemm_nt(a, w, bias, epi=ops.EPI_ACT)
    if epi == "res": return ops.gemm_nt(a, w, bias, epi=ops.EPI_ADD, aux=aux)
    if epi == "dact": return ops.gemm_nt(a, w, epi=ops.EPI_DACT, aux=aux)
    if epi == "none": return ops.gemm_nt(a, w)
    return ops.gemm_nt(a, w, bias)


def race_screen():
    bad = 0
    for (M, N, K) in [(256 * 40 + 77, 1024, 1024), (65536, 4096, 1024), (50000, 768, 3072), (4096, 2304, 96)]:
        torch.manual_seed(M + N)
        a = torch.randn(M, K, device=DEV).to(bf16)
        w = (torch.randn(N, K, device=DEV) * 0.05).to(bf16)
        bias = torch.randn(N, device=DEV)
        aux = torch.randn(M, N, device=DEV).to(bf16)
        for epi in ("bias", "gelu+pre", "res", "dact"):
            lib.debug_set(11, 0)
            ref = call(epi, a, w, bias, aux)
            ref = [t.clone() for t in (ref if isinstance(ref, tuple) else (ref,))]
            for v in (0,):
                for rep in range(6):
                    lib.debug_set(v, 0)
                    got = call(epi, a, w, bias, aux)
                    got = got if isinstance(got, tuple) else (got,)
                    for i, (x, y) in enumerate(zip(ref, got)):
                        if not torch.equal(x, y):
                            bad += 1
                            d = (x.float() - y.float()).abs()
                            emit(kind="MISMATCH", M=M, N=N, K=K, epi=epi, variant=v, rep=rep, out=i,
                                 n_bad=int((d > 0).sum()), max_err=float(d.max()))
    lib.debug_set(0, 0)
    emit(kind="race_screen", mismatches=bad)
    return bad


def bench(M, N, K, epis, variants, rounds, iters):
    torch.manual_seed(1)
    a = torch.randn(M, K, device=DEV).to(bf16)
    w = (torch.randn(N, K, device=DEV) * 0.05).to(bf16)
    bias = torch.randn(N, device=DEV)
    aux = torch.randn(M, N, device=DEV).to(bf16) if any(e in ("res", "dact") for e in epis) else None
    for epi in epis:
        times = {v: [] for v in variants}
        for rnd in range(rounds):
            for v in variants:
                lib.debug_set(v[0], v[1])
                call(epi, a, w, bias, aux)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    call(epi, a, w, bias, aux)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / iters)
        for v in variants:
            t = sorted(times[v])
            med = t[len(t) // 2]
            emit(kind="bench", M=M, N=N, K=K, epi=epi, nt=v[0], abl=v[1], ms_med=round(med, 4), ms_min=round(t[0], 4),
                 tflops_med=round(2 * M * N * K / med / 1e9, 1), tflops_best=round(2 * M * N * K / t[0] / 1e9, 1))
    lib.debug_set(0, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--no-screen", action="store_true")
    ap.add_argument("--exp", default="")
    args = ap.parse_args()
    if not args.no_screen and race_screen():
        sys.exit(1)
    rounds, iters = (3, 2) if args.quick else (5, 3)
    M = 65536 if args.quick else 806912
    V = [(0, 0)]
    VA = V + [(0, 2)]
    bench(M, 4096, 1024, ["gelu+pre", "gelu", "dact", "bias"], VA, rounds, iters)
    bench(M, 1024, 4096, ["res", "none"], VA, rounds, iters)
    bench(M, 3072, 1024, ["bias"], V, rounds, iters)
    bench(M, 1024, 1024, ["res", "none"], V, rounds, iters)
    bench(M, 1024, 3072, ["none"], V, rounds, iters)
    Mt = 32768 if args.quick else 315392
    bench(Mt, 3072, 768, ["gelu+pre", "dact"], V, rounds, iters)
    bench(Mt, 768, 3072, ["res"], V, rounds, iters)
    bench(Mt, 2304, 768, ["bias"], V, rounds, iters)
    bench(Mt, 768, 768, ["res"], V, rounds, iters)


if __name__ == "__main__":
    main()
