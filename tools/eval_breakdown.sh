# Kernel-by-kernel breakdown of the multi-view test path (rocprofv3 kernel trace of tools/bench_eval.py, per batch):
#   bash tools/eval_breakdown.sh [bench_eval args]      (GPU box; stdout)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gev
IT=20; WU=4
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gev -- python tools/bench_eval.py --iters $IT --warmup $WU "$@" > gpurun_out/gev.log 2>&1
python - $IT $WU <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
it, wu = int(sys.argv[1]), int(sys.argv[2])
f = max(glob.glob("gpurun_out/gev/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the timed iterations are the last `it` of it + wu equal batches
n = len(rows) // (it + wu)
seg = rows[-n * it:]
agg = defaultdict(lambda: [0, 0])
for r in seg:
    k = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"]).split("(")[0][:70]
    agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
print("per batch: %.1f us of kernels in %d dispatches, span %.1f us" % (tot / it / 1e3, n, span / it / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%8.1f us %4d  %5.1f%%  %s" % (v[0] / it / 1e3, v[1] // it, 100 * v[0] / tot, k))
PY
rm -rf gpurun_out/gev
