#!/usr/bin/env python
"""Derived metrics from a PMC summary (tools/pmc_summary.py) + the kernel-trace statistics of the same command
(tools/rocpd_stats.py):    python tools/pmc_derive.py gpurun_out/pmcbench_summary.txt gpurun_out/kernel_stats.csv

Per kernel (mean per dispatch): fabric-side bytes (FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note, WRITE_SIZE as counted),
L2 hit rate (TCC_HIT / TCC_REQ), MFMA pipe busy share (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)), the
wave-cycle split (SQ_WAIT_INST_ANY, SQ_WAIT_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES), LDS bank-conflict cycles over LDS-active
cycles, and the effective clock GRBM_GUI_ACTIVE / 8 / duration with the duration of the UNCOUNTED kernel-trace run (an upper
bound on the clock: counters slow a kernel by a few percent)."""
import csv
import re
import sys

summ, stats = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None
per, cur = {}, None
for line in open(summ):
    if line and not line[0].isspace():
        cur = line.strip()
        continue
    m = re.match(r"\s+(\S+)\s+n=\s*(\d+) mean/dispatch=(\S+)", line)
    if m and cur:
        per.setdefault(cur, {})[m.group(1)] = float(m.group(3))
        per[cur]["n"] = int(m.group(2))
dur = {}
if stats:
    for row in csv.DictReader(l for l in open(stats) if not l.startswith("#")):
        dur[row["kernel"][:70]] = float(row["avg_us"])
for k, c in per.items():
    if not any(s in k for s in ("gemm_nta", "gemm_tna", "gemm_tn8", "gemm_f8a", "attn_", "ln_fwd_kernel", "ln_bwd_kernel", "ln_fwd_q8", "quantize_rows")):
        continue
    out = [f"n={c['n']}"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out.append(f"fabric read {2 * c['FETCH_SIZE'] * 1024 / 1e9:.2f} GB (FETCH_SIZE x 2) + write {c['WRITE_SIZE'] * 1024 / 1e9:.2f} GB")
    if c.get("TCC_REQ_sum"):
        out.append(f"L2 hit {100 * c['TCC_HIT_sum'] / c['TCC_REQ_sum']:.1f} %")
    if c.get("GRBM_GUI_ACTIVE"):
        xcd = c["GRBM_GUI_ACTIVE"] / 8
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            out.append(f"MFMA busy {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * xcd):.1f} % of SIMD cycles")
        d = None if " grid=" in k else next((v for kk, v in dur.items() if kk.startswith(k[:60]) or k.startswith(kk[:60])), None)
        if d:      # (rows split by launch grid - tools/pmc_summary.py PMC_BY_GRID - have no duration of their own in the trace summary)
            out.append(f"duration {d:.0f} us (uncounted run) -> clock <= {xcd / d / 1e3:.2f} GHz")
    if c.get("SQ_WAVE_CYCLES"):
        w = c["SQ_WAVE_CYCLES"]
        out.append(f"wave cycles: issue-stalled {100 * c.get('SQ_WAIT_INST_ANY', 0) / w:.1f} %, waiting {100 * c.get('SQ_WAIT_ANY', 0) / w:.1f} %, "
                   f"issuing {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / w:.1f} %")
    if c.get("SQ_LDS_IDX_ACTIVE"):
        out.append(f"LDS bank conflicts {100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.2f} % of LDS-active cycles")
    print(k)
    for o in out:
        print("   ", o)
