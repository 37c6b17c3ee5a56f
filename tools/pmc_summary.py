"""Summarise rocprofv3 --pmc results (rocpd sqlite databases under a directory tree): per kernel, per counter,
the mean value per dispatch.   python tools/pmc_summary.py gpurun_out/pmc [kernel-substring ...]"""
import collections
import glob
import os
import re
import sqlite3
import sys

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
    q = (f"select s.kernel_name, i.name, e.value, d.dispatch_id from {ev} e join {info} i on e.pmc_id = i.id "
         f"join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id")
    per = collections.defaultdict(float)
    for k, c, v, did in cur.execute(q):
        per[(k, c, did)] += float(v)          # sum over instances (XCDs / SEs) of one dispatch
    for (k, c, did), v in per.items():
        acc[re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", k)[:70]][c].append(v)
for k, ctrs in sorted(acc.items()):
    if not any(f in k for f in filters):
        continue
    print(k)
    for c, vals in sorted(ctrs.items()):
        print(f"   {c:34s} n={len(vals):3d} mean/dispatch={sum(vals) / len(vals):.6g}")
