"""Run-length encoded kernel sequence of the tail of a rocprofv3 kernel trace (debug aid).
    python tools/trace_seq.py <kernel_trace.csv> [n_last]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -800:]
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
prev, cnt, dur = None, 0, 0
for s, e, n in rows:
    k = short(n)
    if k == prev:
        cnt += 1; dur += e - s
    else:
        if prev is not None:
            print("%4d x %8.1f us  %s" % (cnt, dur / 1e3, prev))
        prev, cnt, dur = k, 1, e - s
print("%4d x %8.1f us  %s" % (cnt, dur / 1e3, prev))
