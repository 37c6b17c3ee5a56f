"""Throughput of clipa_gemm_nt per epilogue on the ViT-L/16 block shapes (M = 4096 * 49 rows).   python tools/gemm_epi_bench.py [f8]"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops  # noqa: E402

f8 = len(sys.argv) > 1 and sys.argv[1] == "f8"
dev = "cuda"
M = 200704
for N, K in ((4096, 1024), (1024, 4096), (1024, 1024), (3072, 1024), (5120, 1280), (1280, 5120)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16)
    if f8:
        qa, sa = ops.quantize_rows(a)
        qb, sb = ops.quantize_rows(b)
    row = {"kernel": "gemm_nt_f8" if f8 else "gemm_nt", "M": M, "N": N, "K": K}
    for name in ("none", "bias", "gelu", "gelu+pre", "add", "dact"):
        def once():
            kw = dict(epi={"none": ops.EPI_NONE, "bias": ops.EPI_NONE, "gelu": ops.EPI_ACT, "gelu+pre": ops.EPI_ACT, "add": ops.EPI_ADD,
                           "dact": ops.EPI_DACT}[name])
            if name == "gelu+pre":
                kw["want_pre"] = True
            if name in ("add", "dact"):
                kw["aux"] = aux
            bb = None if name in ("none", "dact") else bias
            return ops.gemm_nt_f8(qa, sa, qb, sb, bb, **kw) if f8 else ops.gemm_nt(a, b, bb, **kw)
        once()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                once()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3)
        row[name] = round(2.0 * M * N * K / statistics.median(ts) / 1e9, 1)
    print(json.dumps(row), flush=True)
