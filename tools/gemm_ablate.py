"""Ablation of clipa_gemm_nt's memory behaviour (no numerics checked here): the same launch with the operand loads made
out-of-bounds (lda/ldb = 0 -> the buffer descriptors cover 0 bytes, LDS-DMA writes zeros, nothing is read) and with the
output rows aliased (ldc = 0 -> every row stores to the same 8 KiB, which stays in L2).  What remains is the in-CU schedule.
    python tools/gemm_ablate.py [M N K]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import lib, ops  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (200704, 4096, 1024)
dev = "cuda"
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
bias = torch.randn(N, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run(lda, ldb, ldc, epi_bias=True, iters=10):
    def once():
        lib.call("clipa_gemm_nt", a.data_ptr(), b.data_ptr(), c.data_ptr(), None, bias.data_ptr() if epi_bias else None, None,
                 M, N, K, lda, ldb, ldc, 0, 1.0, 0, 0, 0, st)
    for _ in range(3):
        once()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        once()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9


for name, lda, ldb, ldc in (("normal", K, K, N), ("A_not_read", 0, K, N), ("A_B_not_read", 0, 0, N), ("C_rows_aliased", K, K, 0),
                            ("no_reads_C_aliased", 0, 0, 0), ("normal_again", K, K, N)):
    ms, tf = run(lda, ldb, ldc)
    print(json.dumps({"case": name, "M": M, "N": N, "K": K, "ms": round(ms, 4), "TFLOPs": round(tf, 1)}), flush=True)
