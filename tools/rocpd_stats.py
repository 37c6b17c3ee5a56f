"""Per-kernel statistics (calls, total / average duration, share) from a rocprofv3 rocpd sqlite database
(`rocprofv3 --kernel-trace --stats -o NAME`).   python tools/rocpd_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
                   f"max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
                   f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)


def short(name):
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
    name = re.sub(r"void at::native::", "at::", name)
    return name[:90]


lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,pct,vgpr,agpr,sgpr,lds_bytes"]
for name, n, tot, mn, mx, vg, ag, sg, lds in rows:
    lines.append(f"\"{short(name)}\",{n},{tot / 1e6:.3f},{tot / n / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},"
                 f"{100.0 * tot / total:.2f},{vg},{ag},{sg},{lds}")
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
print("\n".join(lines[:28]))
print(f"# total kernel time {total / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches")
