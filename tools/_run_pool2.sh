cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in 1 100 160 224; do
for b in 4 14 3 2 1; do
rm -rf gpurun_out/p1_tr
SVIT_WGRAD_TILES=$w rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p1_tr -- python tools/pool_one.py $b > gpurun_out/p1_tr.log 2>&1
python - $b $w <<'PY'
import csv, glob, os, re, sys, collections
f=max(glob.glob("gpurun_out/p1_tr/*/*_kernel_trace.csv"), key=os.path.getmtime)
agg=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "wgrad" not in n and "reduce" not in n: continue
    m=re.search(r"(\w+_kernel)", n)
    agg.setdefault((m.group(1), r["Grid_Size_X"]), []).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("want", sys.argv[2], "blk", sys.argv[1], "  ".join("%s g%s %.1f" % (k[0][:22], k[1], sorted(t)[len(t)//2]) for k,t in agg.items()))
PY
done
done
