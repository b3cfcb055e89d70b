"""Averages rocprofv3 --pmc counter CSVs per (kernel, counter)."""
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"{out}/pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
with open(f"{out}/summary.txt", "w") as fh:
    for (kern, ctr), (tot, n) in sorted(acc.items()):
        line = f"{kern:42s} {ctr:36s} avg/dispatch {tot/n:16.1f}  (n={n})"
        print(line); fh.write(line + "\n")
