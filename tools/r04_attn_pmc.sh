# SQ counter passes over the attention forward / dq / dkv kernels at one block shape (GPU box):
#   bash tools/r04_attn_pmc.sh <blk> [tag]
# Counters only with --kernel-trace (gpurun refuses --pmc beside the hip/hsa trace domains); program directly after `--`.
set -e
BLK=$1; TAG=${2:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_BUSY_CU_CYCLES SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_attn_$TAG$i
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/gpurun_out/pmc_attn_$TAG$i -- python3 $R/tools/attn_one.py $BLK bwd > $R/gpurun_out/pmc_attn_$TAG$i.log 2>&1 || echo "pass $i failed (see gpurun_out/pmc_attn_$TAG$i.log)"
done
cd $R
python3 tools/pmc_sq.py $(ls gpurun_out/pmc_attn_$TAG?/*/*counter_collection.csv) --match attn_ > gpurun_out/${TAG}_attn_pmc_blk$BLK.txt
rm -rf gpurun_out/pmc_attn_$TAG?
cat gpurun_out/${TAG}_attn_pmc_blk$BLK.txt
