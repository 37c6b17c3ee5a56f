"""Summarise rocprofv3 --pmc results (rocpd sqlite databases under a directory tree): per kernel, per counter,
the mean value per dispatch.   python tools/pmc_summary.py gpurun_out/pmc [kernel-substring ...]

Kernels whose name contains one of PMC_BY_GRID (environment, comma-separated substrings; default "gemm_tna") are additionally split
by launch grid ("... grid=GXxGY"): one instantiation serves every weight-gradient shape, and its grid (tiles x slices) tells them
apart (per-shape FETCH / WRITE of gemm_tna)."""
import collections
import glob
import os
import re
import sqlite3
import sys

BY_GRID = [t for t in os.environ.get("PMC_BY_GRID", "gemm_tna").split(",") if t]
root = sys.argv[1]
filters = sys.argv[2:] or ["gemm", "attn", "ln_", "colsum"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*.db"), recursive=True):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]

    def tab(prefix):
        t = [x for x in tabs if x.startswith(prefix)]
        return t[0] if t else None

    ev, info, disp, sym = tab("rocpd_pmc_event"), tab("rocpd_info_pmc"), tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
    if not all((ev, info, disp, sym)):
        continue
    dcols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    grid = "d.grid_size_x, d.grid_size_y" if "grid_size_x" in dcols and "grid_size_y" in dcols else "0, 0"
    q = (f"select s.kernel_name, i.name, e.value, d.dispatch_id, {grid} from {ev} e join {info} i on e.pmc_id = i.id "
         f"join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id")
    per = collections.defaultdict(float)
    grids = {}
    for k, c, v, did, gx, gy in cur.execute(q):
        per[(k, c, did)] += float(v)          # sum over instances (XCDs / SEs) of one dispatch
        grids[(k, did)] = (gx, gy)
    for (k, c, did), v in per.items():
        name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", k)[:70]
        acc[name][c].append(v)
        if any(t in k for t in BY_GRID) and grids.get((k, did), (0, 0)) != (0, 0):
            acc[f"{name} grid={grids[(k, did)][0]}x{grids[(k, did)][1]}"][c].append(v)
for k, ctrs in sorted(acc.items()):
    if not any(f in k for f in filters):
        continue
    print(k)
    for c, vals in sorted(ctrs.items()):
        print(f"   {c:34s} n={len(vals):3d} mean/dispatch={sum(vals) / len(vals):.6g}")
