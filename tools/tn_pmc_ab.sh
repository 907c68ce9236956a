# L2 (TCC) request / hit / miss counters, fabric bytes and kernel time of the grouped TN weight-gradient launches INSIDE one eager
# training step, product against a knob arm (round 6: tn_tile=4 = 256x192 tiles on 8-wave workgroups):
#     bash tools/tn_pmc_ab.sh "--set tn_tile=4" > gpurun_out/r06_tn_wide_counters.txt        (GPU box)
# (counters only with --kernel-trace, one pass per counter set, the program directly after `--`)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARM="$1"
for arm in product knob; do
  K=""; [ $arm = knob ] && K="$ARM"
  rm -rf $R/gpurun_out/pmc_tn1 $R/gpurun_out/pmc_tn2
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/pmc_tn1 -- python3 $R/tools/bench_knobs.py $K -- --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-trace > $R/gpurun_out/pmc_tn1.log 2>&1 || echo "pass 1 failed"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_tn2 -- python3 $R/tools/bench_knobs.py $K -- --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-trace > $R/gpurun_out/pmc_tn2.log 2>&1 || echo "pass 2 failed"
  cd $R
  echo "## arm: $arm $K"
  python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
dur = {}
for d in ("pmc_tn1", "pmc_tn2"):
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "gemm_tn_grouped" in r["Kernel_Name"]:
                rows[(d, int(r["Dispatch_Id"]))][r["Counter_Name"]] = float(r["Counter_Value"])
    for f in glob.glob("gpurun_out/%s/*/*kernel_trace.csv" % d):
        for r in csv.DictReader(open(f)):
            if "gemm_tn_grouped" in r["Kernel_Name"]:
                dur[(d, int(r["Dispatch_Id"]))] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
def last8(d):
    ids = sorted(k[1] for k in rows if k[0] == d)[-8:]
    return [(rows[(d, i)], dur.get((d, i), 0.0)) for i in ids]
a, b = last8("pmc_tn1"), last8("pmc_tn2")
tot = [0.0] * 5
for i, ((x, t1), (y, t2)) in enumerate(zip(a, b)):
    req, hit, miss, ea = x.get("TCC_REQ_sum", 0), x.get("TCC_HIT_sum", 0), x.get("TCC_MISS_sum", 0), x.get("TCC_EA0_RDREQ_sum", 0)
    fetch = 2 * y.get("FETCH_SIZE", 0) / 1024.0
    print("launch %d: %7.1f us  L2 req %.3g  hit %.3g (%.1f %%)  miss %.3g  fabric read req %.3g  FETCH %.1f MB"
          % (i, t2, req, hit, 100.0 * hit / max(req, 1), miss, ea, fetch))
    for j, v in enumerate((t2, req, hit, miss, fetch)):
        tot[j] += v
print("all 8   : %7.1f us  L2 req %.3g  hit %.3g (%.1f %%)  miss %.3g  FETCH %.1f MB" % (tot[0], tot[1], tot[2], 100.0 * tot[2] / max(tot[1], 1), tot[3], tot[4]))
PY
  cd /tmp
done
rm -rf $R/gpurun_out/pmc_tn1 $R/gpurun_out/pmc_tn2
