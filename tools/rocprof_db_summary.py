#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace results database (rocprofv3 7.x writes <pid>_results.db):

    python tools/rocprof_db_summary.py gpurun_out/prof_train [--top 30] [--steps N]   (times are then per step)
"""
import glob
import re
import sqlite3
import sys

path = sys.argv[1]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 30
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 1
db = path if path.endswith(".db") else sorted(glob.glob(path + "/**/*_results.db", recursive=True))[-1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from {kt} d join {ks} s "
                      "on d.kernel_id = s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"# {db}: {sum(r[1] for r in rows)} dispatches, {tot / 1e3:.2f} ms of kernel time" + (f" = {tot / 1e3 / steps:.2f} ms per step" if steps > 1 else ""))
print(f"{'share':>6} {'calls':>7} {'total ms':>10} {'avg us':>10}  kernel")
for name, n, t, avg in rows[:top]:
    name = re.sub(r"^_ZN\d+eect?\d+", "", name)
    print(f"{t / tot * 100:5.1f}% {n:7d} {t / 1e3:10.3f} {avg:10.1f}  {name[:100]}")
