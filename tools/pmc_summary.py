"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "?")
            if len(k) > 60:
                k = k[:60]
            try:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            except (KeyError, ValueError):
                pass
for k, ctrs in sorted(acc.items()):
    if not any(s in k for s in ("gemm", "attn", "ln_", "colsum")):
        continue
    print(k)
    for c, vals in sorted(ctrs.items()):
        print(f"   {c:36s} n={len(vals):4d} mean={sum(vals) / len(vals):.6g}")
