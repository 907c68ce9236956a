#!/usr/bin/env python3
"""Two kernel traces of the same replayed step (A = NT GEMM heuristic without the ring kernels, B = with): the NT
GEMM launches are matched by their POSITION in the step (the schedule is identical), so every call is compared
with itself in its real surroundings (cold / warm caches, neighbours), not in an isolated loop.
    python tools/nt_by_position.py <traceA dir> <traceB dir>"""
import csv, glob, os, re, sys
from collections import defaultdict


def last_step(d):
    f = max(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"] and r["Grid_Size_X"] == "92416"]
    return rows[idx[-2] + 1:idx[-1] + 1]


def short(n):
    m = re.search(r"gemm_nt_(v2|ring)_kernel<([^>]*)>", n)
    return ("%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))) if m else None


a, b = last_step(sys.argv[1]), last_step(sys.argv[2])
na = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"]) for r in a if "gemm_nt" in r["Kernel_Name"]]
nb = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"]) for r in b if "gemm_nt" in r["Kernel_Name"]]
assert len(na) == len(nb), (len(na), len(nb))
tot = lambda rows: sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows) / 1e3
print("step: A %.1f us, B %.1f us; NT: A %.1f us, B %.1f us (%d calls)" % (tot(a), tot(b), sum(x[1] for x in na), sum(x[1] for x in nb), len(na)))
agg = defaultdict(lambda: [0, 0.0, 0.0])
for (ka, ta, ga), (kb, tb, gb) in zip(na, nb):
    k = (ka, ga, kb)
    agg[k][0] += 1; agg[k][1] += ta; agg[k][2] += tb
print("%-28s %8s  -> %-28s %5s %9s %9s %8s" % ("A kernel", "A grid", "B kernel", "calls", "A us", "B us", "B-A"))
for (ka, ga, kb), (n, ta, tb) in sorted(agg.items(), key=lambda kv: kv[1][1] - kv[1][2]):
    print("%-28s %8s  -> %-28s %5d %9.1f %9.1f %+8.1f" % (ka, ga, kb, n, ta, tb, tb - ta))
