#!/bin/bash
# Collects PMC counters for one kernel family, one rocprofv3 pass per counter group (separate
# passes: gpurun refuses --pmc combined with trace domains, and slots are limited per block).
# usage: tools/pmc_passes.sh <kernel-regex> <outdir> [bench args...]
set -u
REGEX=$1; OUT=$2; shift 2
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-include-regex "$REGEX" --output-format csv -d "$OUT/pass$i" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-modes "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py "$OUT" ${PMC_GRID:-131072}   # the headline shape only: 256 workgroups x 512 threads
