#!/usr/bin/env python
"""Timing of the HBM-bound kernels of the fp8 step (row quantisers with / without column sums, scaled quantisers, LayerNorm
variants) at production row counts, one JSON line per kernel; `--lib` times a variant build (tools/build_variant.sh).

    python tools/stream8_bench.py [--rows 526336] [--D 1280] [--lib clipa_amd/lib/libclipa_var_X.so]
"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=526336)
ap.add_argument("--D", type=int, default=1280)
ap.add_argument("--lib", default=None)
ap.add_argument("--only", default="")
args = ap.parse_args()
from clipa_amd import lib  # noqa: E402
if args.lib:
    lib.LIB_PATH = os.path.abspath(args.lib)
from clipa_amd import ops  # noqa: E402

dev, bf16 = "cuda", torch.bfloat16
M, D = args.rows, args.D
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, D, device=dev, generator=g).to(bf16)
dy = torch.randn(M, D, device=dev, generator=g).to(bf16)
h = torch.randn(M, 4 * D, device=dev, generator=g).to(bf16)
q3 = torch.randn(M, 3 * D, device=dev, generator=g).to(bf16)
h8 = ops.cast_e4m3(h)
gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
ds = torch.rand(M, device=dev) + 0.5
t = torch.ones(1, device=dev) * 4.0


def timed(fn, reps=6, rounds=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)


cases = {
    "quantize_rows_D": (lambda: ops.quantize_rows(x), 3 * M * D),
    "quantize_rows_D_colsum": (lambda: ops.quantize_rows(x, want_colsum=True), 3 * M * D),
    "quantize_rows_3D_colsum": (lambda: ops.quantize_rows(q3, want_colsum=True), 9 * M * D),
    "quantize_rows_4D": (lambda: ops.quantize_rows(h), 12 * M * D),
    "quantize_rows_4D_colsum": (lambda: ops.quantize_rows(h, want_colsum=True), 12 * M * D),
    "scale_quantize_rows_D": (lambda: ops.scale_quantize_rows(x, ds, t), 3 * M * D),
    "scale_quantize_rows_4D_gelu": (lambda: ops.scale_quantize_rows(h, ds, t, act=0), 12 * M * D),
    "scale_quantize_rows_4D_gelu_in8": (lambda: ops.scale_quantize_rows(h8, ds, t, act=0), 8 * M * D),
    "activation_fwd_e4m3_4D": (lambda: ops.activation_fwd(h8, 0), 12 * M * D),
    "ln_fwd_q8": (lambda: ops.layernorm_fwd_q8(x, gam, bet), 3 * M * D),
    "ln_fwd_q8s": (lambda: ops.layernorm_fwd_q8s(x, gam, bet, ds, t), 3 * M * D),
    "ln_fwd": (lambda: ops.layernorm_fwd(x, gam, bet), 4 * M * D),
    "ln_bwd_dres": (lambda: ops.layernorm_bwd(x, gam, dy, dres=dy), 8 * M * D),
    "ln_bwd_dres_q8": (lambda: ops.layernorm_bwd(x, gam, dy, dres=dy, q8_fmt=0, want_rownorm=True), 9 * M * D),
    "quantize_rows_colsum_D": (lambda: ops.quantize_rows(dy, 0, want_colsum=True, want_rownorm=True), 3 * M * D),
}
for name, (fn, nbytes) in cases.items():
    if args.only and not any(o in name for o in args.only.split(",")):
        continue
    ms = timed(fn)
    print(json.dumps({"kernel": name, "rows": M, "D": D, "ms": round(ms, 4), "gbps": round(nbytes / ms * 1e-6), "lib": os.path.basename(lib.LIB_PATH)}), flush=True)
