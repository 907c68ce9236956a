# per-launch durations of ALL kernels of ONE replayed step, in launch order (rocprofv3 kernel trace of bench.py):
#   bash tools/step_listing.sh [extra bench.py args]    (GPU box; stdout)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gsl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gsl -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-trace "$@" > gpurun_out/gsl.log 2>&1
python - <<'PY'
import csv, glob, os, re
f = max(glob.glob("gpurun_out/gsl/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
ends = [i for i, j in zip(adam, adam[1:] + [None]) if j is None or j != i + 1]
seg = rows[ends[-2] + 1:ends[-1] + 1]
for r in seg:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
    if True:
        print("%8.1f us  %s  grid %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n, r.get("Grid_Size_X", "")))
PY
rm -rf gpurun_out/gsl
