cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 1 2; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-trace --nt-cfg 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('v2   ms/step', d['ms_per_step'])"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-trace 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('ring ms/step', d['ms_per_step'])"
done
rm -rf gpurun_out/gp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gp -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-trace > gpurun_out/gp.log 2>&1
python - <<'PY'
import csv, glob, os, re
from collections import defaultdict
f=max(glob.glob("gpurun_out/gp/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "adamw" in r["Kernel_Name"] and r["Grid_Size_X"]=="92416"]
seg=rows[idx[-2]+1:idx[-1]+1]
agg=defaultdict(lambda:[0,0])
for r in seg:
    n=re.sub(r"\(anonymous namespace\)::|void |at::native::","",r["Kernel_Name"]).split("(")[0][:70]
    agg[n][0]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); agg[n][1]+=1
tot=sum(v[0] for v in agg.values())
print("step total %.1f us, %d dispatches"%(tot/1e3,len(seg)))
nt=sum(v[0] for k,v in agg.items() if "gemm_nt" in k)
print("NT total %.1f us"%(nt/1e3))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][0])[:45]:
    print("%8.1f us %4d  %5.1f%%  %s"%(v[0]/1e3,v[1],100*v[0]/tot,k))
PY
