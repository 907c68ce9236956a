# Kernel-by-kernel breakdown of ONE replayed training step (rocprofv3 kernel trace of bench.py, last step):
#   bash tools/step_breakdown.sh [extra bench.py args]    (on the GPU box; output on stdout)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gp -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-trace "$@" > gpurun_out/gp.log 2>&1
python - <<'PY'
import csv, glob, os, re
from collections import defaultdict
f=max(glob.glob("gpurun_out/gp/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
adam=[i for i,r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
ends=[i for i,j in zip(adam, adam[1:]+[None]) if j is None or j!=i+1]
seg=rows[ends[-2]+1:ends[-1]+1]
agg=defaultdict(lambda:[0,0])
fam=defaultdict(float)
def family(n):
    for k,v in (("gemm_nt","NT GEMM"),("gemm_tn","TN GEMM"),("attn_","attention"),("relq","rel-pos plumbing"),("pool_","pooling"),("ln_","LayerNorm"),("adamw","optimizer"),("sumsq","optimizer"),("reduce_partials","reductions")):
        if k in n: return v
    return "other"
for r in seg:
    n=re.sub(r"\(anonymous namespace\)::|void |at::native::","",r["Kernel_Name"]).split("(")[0][:80]
    d=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    agg[n][0]+=d; agg[n][1]+=1; fam[family(n)]+=d
tot=sum(v[0] for v in agg.values())
print("one replayed step: %.1f us in %d dispatches (span %.1f us)"%(tot/1e3,len(seg),(int(seg[-1]["End_Timestamp"])-int(seg[0]["Start_Timestamp"]))/1e3))
for k,v in sorted(fam.items(), key=lambda kv:-kv[1]): print("  %-18s %8.1f us  %5.1f%%"%(k,v/1e3,100*v/tot))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][0]):
    print("%8.1f us %4d  %5.1f%%  %s"%(v[0]/1e3,v[1],100*v[0]/tot,k))
PY
rm -rf gpurun_out/gp
