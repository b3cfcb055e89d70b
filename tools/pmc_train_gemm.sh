set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_train
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-include-regex "gemm_kernel" --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/tools/train_gemm_bench.py > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $GRAFT_REPO_ROOT; python3 tools/pmc_summary.py $OUT > /dev/null; tail -n 40 $OUT/summary.txt
