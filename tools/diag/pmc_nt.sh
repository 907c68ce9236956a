# PMC passes over one NT GEMM shape (run on the GPU box): bash tools/diag/pmc_nt.sh M N K [tag]
set -e
M=$1; N=$2; K=$3; TAG=${4:-nt}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_INSTS_MFMA" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr GRBM_GUI_ACTIVE" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_$TAG$i
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/gpurun_out/pmc_$TAG$i -- python3 $R/tools/nt_one.py $M $N $K 0 8 > $R/gpurun_out/pmc_$TAG$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python tools/pmc_generic.py $(ls gpurun_out/pmc_$TAG*/*/*counter_collection.csv) --match gemm_nt > gpurun_out/r2_pmc_${TAG}_${M}_${N}_${K}.txt
rm -rf gpurun_out/pmc_$TAG?
cat gpurun_out/r2_pmc_${TAG}_${M}_${N}_${K}.txt
