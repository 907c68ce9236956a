cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sw in 0 1; do
export SVIT_POOL_TILED_SW=$sw
for b in 0 2; do
rm -rf gpurun_out/p1_tr
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p1_tr -- python tools/pool_one.py $b > gpurun_out/p1_tr.log 2>&1
python - $b $sw <<'PY'
import csv, glob, os, re, sys, collections
f=max(glob.glob("gpurun_out/p1_tr/*/*_kernel_trace.csv"), key=os.path.getmtime)
agg=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "pool" not in n: continue
    m=re.search(r"(\w+_kernel)", n)
    agg.setdefault((m.group(1), r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["LDS_Block_Size"]), []).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,t in agg.items():
    t=sorted(t); print("sw", sys.argv[2], "blk", sys.argv[1], "%-24s grid %6s %4s %3s lds %6s  median %7.1f us" % (k+(t[len(t)//2],)))
PY
done
done
