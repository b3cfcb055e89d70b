"""Dispatches and the gaps between them around a step boundary of bench.py's timed loop, from a rocprofv3 results database:

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-modes
    python tools/step_boundary_gaps.py gpurun_out/prof
"""
import glob
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1] + "/*/*_results.db"))[-1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
first = [i for i, r in enumerate(rows) if "stem_conv1" in r[0]][-3]  # the first kernel of a forward, three steps from the end
for i in range(first - 6, first + 4):
    name, start, end, grid = rows[i]
    print(f"{name[:44]:46s} grid {grid:8d} dur {(end - start) / 1000:7.2f} us  gap before {(start - rows[i - 1][2]) / 1000:7.2f} us")
