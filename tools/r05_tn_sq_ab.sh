# SQ counters of the grouped TN GEMM inside one eager training step: the product kernel (128x192 tiles, two workgroups per CU)
# against the three-per-CU arm (tools/diag/variants/tn_three_per_cu.patch built as tools/diag/libsvit_diag_tn3.so with
# -DSVIT_TN_FORCE_MID=1).   bash tools/r05_tn_sq_ab.sh  -> gpurun_out/r05_tn_three_per_cu_sq.txt   (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_tn_three_per_cu_sq.txt
: > $OUT
for ARM in product tn3; do
  if [ $ARM = tn3 ]; then export SVIT_HIP_LIB=$R/tools/diag/libsvit_diag_tn3.so; else unset SVIT_HIP_LIB; fi
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
             "SQ_BUSY_CU_CYCLES SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf $R/gpurun_out/pmc_tn3_$i
    rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/gpurun_out/pmc_tn3_$i -- python3 $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-trace > $R/gpurun_out/pmc_tn3_$i.log 2>&1 || echo "pass $i failed"
    echo "$ARM pass $i done" >> $R/gpurun_out/r05_tn_sq_progress.txt
  done
  echo "## arm: $ARM" >> $OUT
  (cd $R && python3 tools/pmc_sq.py $(ls gpurun_out/pmc_tn3_?/*/*counter_collection.csv) --match gemm_tn_grouped) >> $OUT
  rm -rf $R/gpurun_out/pmc_tn3_?
done
unset SVIT_HIP_LIB
cat $OUT
