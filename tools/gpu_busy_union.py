"""GPU busy time (union of kernel intervals over all streams) against the wall span, and the largest idle gaps, over the
second half of a rocprofv3 kernel trace:  python tools/gpu_busy_union.py <rocprof output dir>"""
import glob
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1] + "/*/*_results.db"))[-1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
rows = rows[len(rows) // 2:]
span = rows[-1][2] - rows[0][1]
busy, gaps = 0, []
cs, ce, prev = rows[0][1], rows[0][2], rows[0][0]
for name, s, e in rows[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append((s - ce, prev, name))
        cs, ce = s, e
    if e >= ce:
        ce, prev = e, name
busy += ce - cs
print(f"{len(rows)} dispatches: span {span / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms ({100.0 * busy / span:.1f} %), kernel time summed {sum(r[2] - r[1] for r in rows) / 1e6:.2f} ms")
gaps.sort(reverse=True)
print(f"idle {sum(g[0] for g in gaps) / 1e6:.2f} ms in {len(gaps)} gaps; {sum(1 for g in gaps if g[0] > 20000)} above 20 us; largest:")
for g, a, b in gaps[:12]:
    print(f"  {g / 1000:8.1f} us after {a[:60]} before {b[:60]}")
