# L2 (TCC) hit / miss / HBM-request counters of the grouped TN weight-gradient launches INSIDE one eager training step
# (GPU box): bash tools/r04_tn_pmc.sh   -> gpurun_out/r04_tn_l2.txt
# (counters only with --kernel-trace; the program directly after `--`)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_tn1 $R/gpurun_out/pmc_tn2
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/pmc_tn1 -- python3 $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-trace > $R/gpurun_out/pmc_tn1.log 2>&1 || echo "pass 1 failed"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_tn2 -- python3 $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-trace > $R/gpurun_out/pmc_tn2.log 2>&1 || echo "pass 2 failed"
cd $R
python3 - <<'PY' > gpurun_out/r04_tn_l2.txt
import csv, glob, collections
rows = collections.defaultdict(dict)
for d in ("pmc_tn1", "pmc_tn2"):
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "gemm_tn_grouped" in r["Kernel_Name"]:
                rows[(d, int(r["Dispatch_Id"]))][r["Counter_Name"]] = float(r["Counter_Value"])
def last8(d):
    ids = sorted(k[1] for k in rows if k[0] == d)[-8:]
    return [rows[(d, i)] for i in ids]
a, b = last8("pmc_tn1"), last8("pmc_tn2")
print("# grouped TN launches of the last eager step, in launch order (blocks 15-14, 13-12, ..., 1-0)")
print("# TCC_REQ = L2 requests (128-B lines), HIT / MISS, EA0_RDREQ = read requests L2 -> fabric (Infinity Cache / HBM; 32 or 64 B each), FETCH_SIZE x2 = bytes from the fabric (KB)")
for i, (x, y) in enumerate(zip(a, b)):
    req, hit, miss, ea = x.get("TCC_REQ_sum", 0), x.get("TCC_HIT_sum", 0), x.get("TCC_MISS_sum", 0), x.get("TCC_EA0_RDREQ_sum", 0)
    print("launch %d: L2 req %.3g  hit %.3g (%.1f %%)  miss %.3g  fabric read req %.3g  FETCH %.1f MB"
          % (i, req, hit, 100.0 * hit / max(req, 1), miss, ea, 2 * y.get("FETCH_SIZE", 0) / 1024.0))
PY
rm -rf gpurun_out/pmc_tn1 gpurun_out/pmc_tn2
cat gpurun_out/r04_tn_l2.txt
