#!/usr/bin/env python
"""profiles/traffic.json from a PMC summary (tools/pmc_summary.py output of `tools/gpu_round5.sh pmcbench`):

    python tools/make_traffic_json.py gpurun_out/pmcbench_summary.txt "what was measured"

bench.py reports roofline.traffic from that file only while the GEMM kernel sources it names (kernel_source_sha16) are the
ones that are built.  Corrections per /opt/skills/guides/MI355X_MICROARCH.md: counters are KB per dispatch; FETCH_SIZE reports
half of wide coalesced reads on gfx950 (doubled here), WRITE_SIZE is taken as is."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha16  # noqa: E402


def main():
    text = open(sys.argv[1]).read()
    note = sys.argv[2] if len(sys.argv) > 2 else ""
    per, cur = {}, None
    for line in text.splitlines():
        if line and not line[0].isspace():
            cur = line.strip()
            continue
        m = re.match(r"\s+(\S+)\s+n=\s*(\d+) mean/dispatch=(\S+)", line)
        if m and cur and m.group(1) in ("FETCH_SIZE", "WRITE_SIZE"):
            per.setdefault(cur, {})[m.group(1)] = float(m.group(3))
            per[cur]["n"] = int(m.group(2))
    per = {k: v for k, v in per.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    nt = {k: v for k, v in per.items() if "gemm_nta_kernel" in k}
    n = sum(v["n"] for v in nt.values())
    fetch = sum(v["FETCH_SIZE"] * v["n"] for v in nt.values()) / n * 1024
    write = sum(v["WRITE_SIZE"] * v["n"] for v in nt.values()) / n * 1024
    out = {
        "gemm_nt_hbm_bytes_per_launch": int(2 * fetch + write),
        "uncorrected_fetch_plus_write_bytes": int(fetch + write),
        "kernel_source_sha16": kernel_source_sha16(),
        "measured": note,
        "workload": {"model": "ViT-L-16", "image_size": 224, "ctx": 77, "batch": 4096, "precision": "bf16"},   # bench.py's default = what the PMC stage runs
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, counters only; tools/gpu_round5.sh stage pmcbench) over "
                  "`python bench.py --steps 1 --warmup 0 --keep-blocks <the plan named under 'measured'>`, dispatch-weighted mean over the %d gemm_nta_kernel dispatches of the "
                  "step; read side doubled per MI355X_MICROARCH.md (FETCH_SIZE reports 1/2 of wide coalesced reads on gfx950), WRITE_SIZE taken as "
                  "is.  Fabric-side counters: Infinity-Cache hits are included; the excess over the algorithmic bytes is operand panels re-fetched "
                  "through the fabric by each XCD's 4 MiB L2, not HBM traffic." % n,
        "per_kernel_KB_per_dispatch": {k: v for k, v in per.items() if "gemm_" in k or "attn_" in k},
    }
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(out["gemm_nt_hbm_bytes_per_launch"], out["kernel_source_sha16"], n)


if __name__ == "__main__":
    main()
