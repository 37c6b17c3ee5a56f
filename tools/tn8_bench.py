#!/usr/bin/env python
"""A/B of the fp8 weight-gradient GEMM (clipa_gemm_tn_f8, every schedule of tools/gen_gemm_tn8.py) against the bf16 one
(clipa_gemm_tn) on the production shapes of ViT-H/14 (local batch 2048) and ViT-L/16 (4096): JSON lines on stdout.

    python tools/tn8_bench.py [--iters 10] [--shapes h14|l16|all]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import lib, ops  # noqa: E402

SHAPES = {
    "h14": [(526336, 5120, 1280), (526336, 1280, 5120), (526336, 3840, 1280), (526336, 1280, 1280),
            (157696, 4096, 1024), (157696, 1024, 4096), (157696, 3072, 1024), (157696, 1024, 1024)],
    "l16": [(806912, 4096, 1024), (806912, 1024, 4096), (806912, 3072, 1024), (806912, 1024, 1024),
            (315392, 3072, 768), (315392, 768, 3072), (315392, 2304, 768), (315392, 768, 768)],
}


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--shapes", default="h14")
    ap.add_argument("--schedules", default="0,1,2")
    ap.add_argument("--orders", default="", help="comma-separated extra work orders to time with the default schedule: 4096 = slice-per-XCD forced, 8192 = tile order forced (clipa_internal_debug_set flags)")
    args = ap.parse_args()
    shapes = SHAPES["h14"] + SHAPES["l16"] if args.shapes == "all" else SHAPES[args.shapes]
    dev = "cuda"
    for M, R, C in shapes:
        g = torch.Generator(device=dev).manual_seed(M + R)
        dy = (torch.randn(M, R, device=dev, generator=g) * 1e-3).to(torch.bfloat16)
        x = torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16)
        dq, ds = ops.quantize_rows(dy)
        _, sx = ops.quantize_rows(x)
        t = ops.rowscale_max(ds, sx)
        x8 = ops.scale_quantize_rows(x, ds, t)
        flops = 2.0 * M * R * C
        rec = {"M": M, "R": R, "C": C}
        ms = timed(lambda: ops.gemm_tn(dy, x, torch.bfloat16), args.iters)
        rec["bf16_ms"], rec["bf16_tflops"] = round(ms, 3), round(flops / ms / 1e9, 1)
        ref = ops.gemm_tn(dy, x, torch.float32)
        for s in [int(v) for v in args.schedules.split(",")]:
            lib.debug_set(0, (s + 1) << 26)
            ms = timed(lambda: ops.gemm_tn_f8(dq, x8, t=t, out_dtype=torch.bfloat16), args.iters)
            out = ops.gemm_tn_f8(dq, x8, t=t)
            rel = ((out - ref).norm() / ref.norm()).item()
            rec[f"f8_s{s}_ms"], rec[f"f8_s{s}_tflops"], rec[f"f8_s{s}_rel_err_vs_bf16"] = round(ms, 3), round(flops / ms / 1e9, 1), round(rel, 4)
        for o in [int(v) for v in args.orders.split(",") if v]:
            lib.debug_set(0, o)
            ms = timed(lambda: ops.gemm_tn_f8(dq, x8, t=t, out_dtype=torch.bfloat16), args.iters)
            rec[f"f8_order{o}_ms"], rec[f"f8_order{o}_tflops"] = round(ms, 3), round(flops / ms / 1e9, 1)
        lib.debug_set(0, 0)
        rec["quantize_x_ms"] = round(timed(lambda: ops.scale_quantize_rows(x, ds, t), args.iters), 3)
        print(json.dumps(rec), flush=True)
        del dy, x, dq, x8


if __name__ == "__main__":
    main()
