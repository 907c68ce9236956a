cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gp -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-trace > gpurun_out/gp.log 2>&1
python - <<'PY'
import csv, glob, os, re
f=max(glob.glob("gpurun_out/gp/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
adam=[i for i,r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
ends=[i for i,j in zip(adam, adam[1:]+[None]) if j is None or j!=i+1]
seg=rows[ends[-2]+1:ends[-1]+1]
def sh(n):
    n=re.sub(r"\(anonymous namespace\)::|void |at::native::","",n).split("(")[0]
    return n[:58]
t0=int(seg[0]["Start_Timestamp"])
for i,r in enumerate(seg):
    print("%4d %9.1f %7.1f  %s"%(i,(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, sh(r["Kernel_Name"])))
PY
rm -rf gpurun_out/gp
