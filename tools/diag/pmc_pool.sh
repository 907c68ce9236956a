# SQ counters of the pooling forward kernels (run on the GPU box): bash tools/diag/pmc_pool.sh h T H W sq skv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for MODE in sel stream; do
  rm -rf $R/gpurun_out/pmc_pool_$MODE
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_pool_$MODE -- python3 $R/tools/diag/pool_one.py $@ $MODE > $R/gpurun_out/pmc_pool_$MODE.log 2>&1 || echo "pass $MODE failed"
  echo "== $MODE" >> $R/gpurun_out/r2_pmc_pool.txt
  python3 $R/tools/pmc_generic.py $(ls $R/gpurun_out/pmc_pool_$MODE/*/*counter_collection.csv) --match pool >> $R/gpurun_out/r2_pmc_pool.txt
  python3 - <<PY >> $R/gpurun_out/r2_pmc_pool.txt
import csv, glob
f = glob.glob("$R/gpurun_out/pmc_pool_$MODE/*/*kernel_trace.csv")[0]
for r in csv.DictReader(open(f)):
    if "pool" in r["Kernel_Name"]:
        print("   ", r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
  rm -rf $R/gpurun_out/pmc_pool_$MODE
done
cat $R/gpurun_out/r2_pmc_pool.txt
