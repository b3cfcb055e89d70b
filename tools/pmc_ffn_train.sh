#!/bin/bash
# PMC passes over the training step's fused feed-forward launches alone (tools/ffn_train_bench_{fwd,bwd}); one pass per counter group
OUT=gpurun_out/r04o/pmc_tr; mkdir -p $OUT; export TMPDIR=/tmp
for b in fwd bwd; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --kernel-include-regex "ffn_chain" --output-format csv -d $OUT/${b}_pass$i -- tools/ffn_train_bench_$b > $OUT/${b}_pass$i.log 2>&1 || echo "$b pass $i failed"
  done
done
python3 - <<'PY'
import csv, glob, collections
for b in ("fwd", "bwd"):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f"gpurun_out/r04o/pmc_tr/{b}_pass*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(b, {k: (v / n[k]) for k, v in tot.items()}, "launches", max(n.values()) if n else 0)
PY
