"""Averages rocprofv3 --pmc counter CSVs per (kernel, counter).

    python tools/pmc_summary.py <outdir> [grid_size]

grid_size (work-items, e.g. 131072 = 256 workgroups x 512 threads): only dispatches of that size are averaged -- bench.py also
launches the same kernels on the small parity fixtures (B = 4), which would drag the averages of the headline shape down."""
import csv, glob, sys, collections
out = sys.argv[1]
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"{out}/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if grid and int(r["Grid_Size"]) != grid:
            continue
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
with open(f"{out}/summary.txt", "w") as fh:
    for (kern, ctr), (tot, n) in sorted(acc.items()):
        line = f"{kern:42s} {ctr:36s} avg/dispatch {tot/n:16.1f}  (n={n})"
        print(line); fh.write(line + "\n")
