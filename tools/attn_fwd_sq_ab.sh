# SQ counters of the attention FORWARD at one block shape, product against the anti-phase variant library
# (tools/diag/libsvit_diag_apfull.so = the tree with tools/diag/variants/attn_fwd_anti_phase.patch applied, tools/diag/build_all_variant.py apfull):
#     bash tools/attn_fwd_sq_ab.sh <blk>  > gpurun_out/r06_attn_fwd_sq_blk<blk>.txt      (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
BLK=$1
for ARM in product ap; do
  if [ $ARM = ap ]; then export SVIT_HIP_LIB=$R/tools/diag/libsvit_diag_apfull.so; X=ap; else unset SVIT_HIP_LIB; X=""; fi
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
             "SQ_BUSY_CU_CYCLES SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf $R/gpurun_out/pmc_af_$i
    rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/gpurun_out/pmc_af_$i -- python3 $R/tools/attn_one.py $BLK $X > $R/gpurun_out/pmc_af_$i.log 2>&1 || echo "pass $i failed"
  done
  echo "## arm: $ARM (blk $BLK)"
  (cd $R && python3 tools/pmc_sq.py $(ls gpurun_out/pmc_af_?/*/*counter_collection.csv) --match attn_fwd)
  rm -rf $R/gpurun_out/pmc_af_?
done
