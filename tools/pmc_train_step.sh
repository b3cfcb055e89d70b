#!/bin/bash
# HBM-side traffic of one training step: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/train_profile.py
# (3 steps), summed per kernel.  usage: bash tools/pmc_train_step.sh   (writes gpurun_out/pmc_train_step/)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_train_step
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -- python3 "$ROOT/tools/train_profile.py" --passes 3 --steps 2 > "$OUT/$c.log" 2>&1 || echo "$c pass failed"
done
cd "$ROOT"
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/pmc_train_step")
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            tot[r["Kernel_Name"].split("(")[0][-60:]][c] += float(r["Counter_Value"])
rows = sorted(tot.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))
steps = 3
gb = lambda kib: kib * 1024 / 1e9
print(f"# HBM-side traffic per training step (default model, B = 64, bf16x3; 3 profiled steps / 3): 2 x FETCH_SIZE + WRITE_SIZE (KiB -> GB)")
s_f = sum(v["FETCH_SIZE"] for v in tot.values()); s_w = sum(v["WRITE_SIZE"] for v in tot.values())
print(f"total: fetch {gb(2 * s_f) / steps:.2f} GB + write {gb(s_w) / steps:.2f} GB = {gb(2 * s_f + s_w) / steps:.2f} GB per step")
for k, v in rows[:16]:
    print(f"{gb(2 * v['FETCH_SIZE']) / steps:8.2f} GB read {gb(v['WRITE_SIZE']) / steps:8.2f} GB written   {k}")
PY
