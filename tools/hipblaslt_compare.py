"""Reference point for the roofline discussion: the vendor library (hipBLASLt / rocBLAS through torch.nn.functional.linear) on the
same bf16 GEMM shapes, same process, interleaved with clipa_gemm_nt.  Not part of the product path.
    python tools/hipblaslt_compare.py"""
import json
import os
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops  # noqa: E402

dev = "cuda"
M = 200704
for N, K in ((4096, 1024), (1024, 4096), (3072, 1024), (1024, 1024), (5120, 1280), (1280, 5120)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias32 = torch.randn(N, device=dev)
    bias16 = bias32.to(torch.bfloat16)
    fns = {
        "clipa_gemm_nt(bias)": lambda: ops.gemm_nt(a, w, bias32),
        "torch_linear(bias)": lambda: F.linear(a, w, bias16),
        "clipa_gemm_nt(no bias)": lambda: ops.gemm_nt(a, w, None),
        "torch_matmul": lambda: torch.matmul(a, w.t()),
    }
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    ts = {k: [] for k in fns}
    for _ in range(5):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) / 3)
    row = {"M": M, "N": N, "K": K}
    for k, t in ts.items():
        row[k] = round(2.0 * M * N * K / statistics.median(t) / 1e9, 1)
    print(json.dumps(row), flush=True)
