#!/usr/bin/env python
"""The e4m3-operand GELU-backward GEMM (clipa_gemm_nt CLIPA_EPI_DACT8 / clipa_gemm_nt_f8 with an e4m3 aux) and its neighbours at
production shapes, one JSON line per shape; `--lib` times another build of the library (A/B of the table-driven epilogue).

    python tools/dact8_bench.py [--lib tools/probes/var/x.so] [--f8]
"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--f8", action="store_true")
args = ap.parse_args()
from clipa_amd import lib  # noqa: E402
if args.lib:
    lib.LIB_PATH = os.path.abspath(args.lib)
from clipa_amd import ops  # noqa: E402

dev = "cuda"
shapes = [(526336, 5120, 1280), (157696, 4096, 1024)] if args.f8 else [(806912, 4096, 1024), (315392, 3072, 768)]


def timed(fn):
    fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    return statistics.median(ts)


for M, N, K in shapes:
    g = torch.Generator(device=dev).manual_seed(M)
    dy = (torch.randn(M, K, device=dev, generator=g) * 1e-3).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) * 0.04).to(torch.bfloat16)
    h8 = ops.cast_e4m3((torch.randn(M, N, device=dev, generator=g) * 1.5).to(torch.bfloat16))
    rec = {"M": M, "N": N, "K": K, "lib": os.path.basename(lib.LIB_PATH)}
    if args.f8:
        dq, ds = ops.quantize_rows(dy)
        wq, ws = ops.quantize_rows(w)
        cases = {"plain": lambda: ops.gemm_nt_f8(dq, ds, wq, ws), "dact8": lambda: ops.gemm_nt_f8(dq, ds, wq, ws, epi=ops.EPI_DACT, aux=h8)}
    else:
        cases = {"plain": lambda: ops.gemm_nt(dy, w), "dact8": lambda: ops.gemm_nt(dy, w, epi=ops.EPI_DACT, aux=h8),
                 "dact8_act": lambda: ops.gemm_nt(dy, w, epi=ops.EPI_DACT, aux=h8, want_act=True)}
    for name, fn in cases.items():
        ms = timed(fn)
        rec[name + "_ms"], rec[name + "_tflops"] = round(ms, 3), round(2.0 * M * N * K / ms / 1e9, 1)
    if not args.f8:
        rec["activation_fwd_e4m3_ms"] = round(timed(lambda: ops.activation_fwd(h8, 0)), 3)
    out = cases["dact8"]()
    rec["checksum"] = float(out.float().abs().sum())
    print(json.dumps(rec), flush=True)
    del dy, w, h8
