"""Per-kernel micro-benchmarks on one MI355X (HIP events, random data). Writes JSON lines.
    python tools/microbench.py [--quick] > gpurun_out/microbench.jsonl
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops  # noqa: E402

DEV = "cuda"
bf16 = torch.bfloat16


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def emit(**kw):
    print(json.dumps(kw), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(0)
    Ms = [65536] if args.quick else [65536, 806912]
    for M in Ms:
        for (N, K, tag) in [(3072, 1024, "L qkv"), (1024, 1024, "L out"), (4096, 1024, "L fc1"), (1024, 4096, "L fc2"),
                            (2304, 768, "B qkv"), (3072, 768, "B fc1")]:
            a = torch.randn(M, K, device=DEV).to(bf16)
            w = (torch.randn(N, K, device=DEV) * 0.05).to(bf16)
            bias = torch.randn(N, device=DEV)
            ms = timeit(lambda: ops.gemm_nt(a, w, bias))
            emit(kernel="gemm_nt", tag=tag, M=M, N=N, K=K, epi="bias", ms=round(ms, 3), tflops=round(2 * M * N * K / ms / 1e9, 1))
            if tag in ("L fc1",):
                ms = timeit(lambda: ops.gemm_nt(a, w, bias, epi=ops.EPI_ACT, want_pre=True))
                emit(kernel="gemm_nt", tag=tag, M=M, N=N, K=K, epi="gelu+pre", ms=round(ms, 3), tflops=round(2 * M * N * K / ms / 1e9, 1))
            if tag in ("L out", "L fc2"):
                res = torch.randn(M, N, device=DEV).to(bf16)
                ms = timeit(lambda: ops.gemm_nt(a, w, bias, epi=ops.EPI_ADD, aux=res))
                emit(kernel="gemm_nt", tag=tag, M=M, N=N, K=K, epi="residual", ms=round(ms, 3), tflops=round(2 * M * N * K / ms / 1e9, 1))
                del res
            # weight gradient of the same layer: dW[N,K] = dY[M,N]^T X[M,K]
            dy = torch.randn(M, N, device=DEV).to(bf16)
            ms = timeit(lambda: ops.gemm_tn(dy, a))
            emit(kernel="gemm_tn", tag=tag, M=M, R=N, C=K, ms=round(ms, 3), tflops=round(2 * M * N * K / ms / 1e9, 1))
            del a, w, dy
    # square reference points (guide: 8192^3)
    for n in (4096, 8192):
        a = torch.randn(n, n, device=DEV).to(bf16)
        b = torch.randn(n, n, device=DEV).to(bf16)
        ms = timeit(lambda: ops.gemm_nt(a, b))
        emit(kernel="gemm_nt", tag="square", M=n, N=n, K=n, ms=round(ms, 3), tflops=round(2 * n ** 3 / ms / 1e9, 1))
        ms = timeit(lambda: ops.gemm_tn(a, b))
        emit(kernel="gemm_tn", tag="square", M=n, R=n, C=n, ms=round(ms, 3), tflops=round(2 * n ** 3 / ms / 1e9, 1))
        del a, b
    for (B, H, L, causal, tag) in [(512, 16, 197, False, "L/16@224"), (512, 12, 77, True, "text-77"), (1024, 16, 26, False, "L/16@84"),
                                   (256, 16, 257, False, "14@224")]:
        D = 64 * H
        qkv = torch.randn(B * L, 3 * D, device=DEV).to(bf16)
        ms = timeit(lambda: ops.attention_fwd(qkv, B, L, H, causal))
        fl = 4.0 * B * H * L * L * 64 * (0.5 if causal else 1)
        emit(kernel="attention_fwd", tag=tag, B=B, H=H, L=L, ms=round(ms, 3), tflops=round(fl / ms / 1e9, 1))
        o, st = ops.attention_fwd(qkv, B, L, H, causal, want_stats=True)
        do = torch.randn_like(o)
        ms = timeit(lambda: ops.attention_bwd(qkv, o, do, st, B, L, H, causal))
        emit(kernel="attention_bwd", tag=tag, B=B, H=H, L=L, ms=round(ms, 3), tflops=round(2.5 * fl / ms / 1e9, 1))
        del qkv, o, do
    for D in (768, 1024):
        rows = 806912 if not args.quick else 131072
        x = torch.randn(rows, D, device=DEV).to(bf16)
        w, b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
        ms = timeit(lambda: ops.layernorm_fwd(x, w, b))
        emit(kernel="layernorm_fwd", rows=rows, D=D, ms=round(ms, 3), gbps=round(2 * rows * D * 2 / ms / 1e6, 1))
        dy = torch.randn_like(x)
        ms = timeit(lambda: ops.layernorm_bwd(x, w, dy, x))
        emit(kernel="layernorm_bwd", rows=rows, D=D, ms=round(ms, 3), gbps=round(4 * rows * D * 2 / ms / 1e6, 1))
        ms = timeit(lambda: ops.colsum(x))
        emit(kernel="colsum", rows=rows, D=D, ms=round(ms, 3), gbps=round(rows * D * 2 / ms / 1e6, 1))
        del x, dy


if __name__ == "__main__":
    main()
