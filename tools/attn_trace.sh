# per-kernel durations of the attention launches per block shape (rocprofv3 kernel trace of tools/attn_one.py)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/attn_tr
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/attn_tr -- python tools/attn_shapes.py > gpurun_out/attn_tr.log 2>&1
K=$(ls gpurun_out/attn_tr/*/*kernel_trace.csv | head -1)
python - "$K" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X"), r.get("Workgroup_Size_X"), r.get("VGPR_Count", ""), r.get("LDS_Block_Size","")) for r in rows if "attn_" in r["Kernel_Name"]]
# group consecutive identical kernel names -> median
out = []
i = 0
while i < len(seq):
    j = i
    while j < len(seq) and seq[j][0] == seq[i][0] and seq[j][2] == seq[i][2]:
        j += 1
    # pattern may alternate (dq, dkv): handle by collecting per (name, grid)
    i = j
agg = collections.OrderedDict()
for n, us, g, w, v, l in seq:
    agg.setdefault((n, g, w, v, l), []).append(us)
for (n, g, w, v, l), t in agg.items():
    t = sorted(t)
    short = n.replace("void (anonymous namespace)::", "").split("(")[0]
    print("%-60s grid %8s wg %4s vgpr %4s lds %6s  n=%3d  median %8.1f us  min %8.1f" % (short[:60], g, w, v, l, len(t), t[len(t)//2], t[0]))
PY
