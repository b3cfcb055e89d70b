set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc_inf; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-modes > $OUT/$c.log 2>&1 || echo "$c failed"
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, os
out=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/pmc_inf")
tot=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob(f"{out}/{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][-58:]; tot[k][c]+=float(r["Counter_Value"])
            if c=="FETCH_SIZE": cnt[k]+=1
gb=lambda kib: kib*1024/1e6
print("# HBM-side traffic per launch (MB; 2 x FETCH_SIZE + WRITE_SIZE, KiB -> B), python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-modes")
for k,v in sorted(tot.items(), key=lambda kv:-(2*kv[1]["FETCH_SIZE"]+kv[1]["WRITE_SIZE"]))[:10]:
    n=max(cnt[k],1); print(f"{gb(2*v['FETCH_SIZE'])/n:9.1f} MB read {gb(v['WRITE_SIZE'])/n:9.1f} MB written  x{n:5d}  {k}")
PY
