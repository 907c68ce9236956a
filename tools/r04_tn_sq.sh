# SQ counter passes over EVERY kernel of one eager training step (GPU box), summarised per family:
#   bash tools/r04_tn_sq.sh   -> gpurun_out/r04_tn_sq.txt (grouped TN), r04_nt_sq.txt (NT GEMMs), r04_pool_sq.txt, r04_ln_sq.txt
# (counters only with --kernel-trace; the program directly after `--`)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_BUSY_CU_CYCLES SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_tnsq$i
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/gpurun_out/pmc_tnsq$i -- python3 $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-trace > $R/gpurun_out/pmc_tnsq$i.log 2>&1 || echo "pass $i failed (see gpurun_out/pmc_tnsq$i.log)"
  echo "pass $i done" >> $R/gpurun_out/r04_tn_sq_progress.txt
done
cd $R
python3 tools/pmc_sq.py $(ls gpurun_out/pmc_tnsq?/*/*counter_collection.csv) --match gemm_tn_grouped > gpurun_out/r04_tn_sq.txt
python3 tools/pmc_sq.py $(ls gpurun_out/pmc_tnsq?/*/*counter_collection.csv) --match gemm_nt_ > gpurun_out/r04_nt_sq.txt
python3 tools/pmc_sq.py $(ls gpurun_out/pmc_tnsq?/*/*counter_collection.csv) --match pool_ > gpurun_out/r04_pool_sq.txt
python3 tools/pmc_sq.py $(ls gpurun_out/pmc_tnsq?/*/*counter_collection.csv) --match ln_ > gpurun_out/r04_ln_sq.txt
rm -rf gpurun_out/pmc_tnsq?
cat gpurun_out/r04_tn_sq.txt
