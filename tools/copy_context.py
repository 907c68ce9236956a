"""Which kernels surround the runtime's D2D copy kernel in a rocprofv3 kernel trace (debug aid).
    python tools/copy_context.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[-4000:]
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
ctx = collections.Counter()
dur = collections.Counter()
for i, (s, e, n) in enumerate(rows):
    if "copyBuffer" in n:
        key = (short(rows[i - 1][2]) if i else "-", short(rows[i + 1][2]) if i + 1 < len(rows) else "-")
        ctx[key] += 1
        dur[key] += e - s
for k, v in ctx.most_common(25):
    print("%4d x  %7.1f us   after %-48s before %s" % (v, dur[k] / 1e3, k[0], k[1]))
