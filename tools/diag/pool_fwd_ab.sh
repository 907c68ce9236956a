# Forward pooling kernels of one block shape under rocprofv3, VALU slab conv (SVIT_POOL_SLAB=1) against the MFMA conv (=2):
#   bash tools/diag/pool_fwd_ab.sh [blk ...]      (GPU box; prints kernel, calls, average us)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for blk in "${@:-4}"; do
  for v in 1 2; do
    rm -rf $R/gpurun_out/pfab
    SVIT_POOL_SLAB=$v rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pfab -- python3 $R/tools/pool_one.py $blk > $R/gpurun_out/pfab.log 2>&1 || { echo "rocprofv3 failed"; tail -5 $R/gpurun_out/pfab.log; }
    echo "== blk $blk SVIT_POOL_SLAB=$v"
    python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/pfab/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "pool_" in n:
        print("   %-60s calls %4s  avg %8.1f us" % (n.replace("(anonymous namespace)::", "").split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
rm -rf $R/gpurun_out/pfab
