# Per-kernel launch geometry of one replayed step: waves per CU a launch offers against what its registers / LDS admit
# (GPU box): bash tools/occupancy_table.sh > gpurun_out/r04_occupancy.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/occ
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/occ -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-trace > gpurun_out/occ.log 2>&1
python - <<'PY'
import csv, glob, os, re
from collections import defaultdict
f = max(glob.glob("gpurun_out/occ/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
ends = [i for i, j in zip(adam, adam[1:] + [None]) if j is None or j != i + 1]
seg = rows[ends[-2] + 1:ends[-1] + 1]
agg = defaultdict(lambda: [0, 0.0, None])
for r in seg:
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"]).split("(")[0][:56]
    gx = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    key = (n, gx, wg, r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "0"), r.get("LDS_Block_Size", "?"))
    agg[key][0] += 1
    agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("%-56s %9s %5s %5s %7s %6s %8s %9s %9s" % ("kernel", "threads", "wg", "vgpr", "lds", "calls", "us/call", "waves/CU", "resident"))
for (n, gx, wg, vg, ag, lds), (c, t, _) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    waves = gx / 64.0 / 256.0
    try:
        regs = int(vg) + int(ag or 0)
        alloc = (regs + 7) // 8 * 8
        by_reg = min(8, 512 // max(alloc, 1)) * 4
        l = int(lds)
        wpw = wg // 64
        by_lds = (160 * 1024 // l) * wpw if l > 0 else 32
        res = min(32, by_reg, by_lds)
    except ValueError:
        res = -1
    print("%-56s %9d %5d %5s %7s %6d %8.1f %9.1f %9d" % (n, gx, wg, vg, lds, c, t / c, waves, res))
PY
rm -rf gpurun_out/occ
