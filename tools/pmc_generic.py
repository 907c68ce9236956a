#!/usr/bin/env python3
"""Per-kernel averages of every counter in rocprofv3 --pmc counter_collection.csv files.
usage: pmc_generic.py <csv> [<csv> ...] [--match substring]"""
import csv, sys, collections
match = None
files = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--match": match = args.pop(0)
    else: files.append(a)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if match and match not in k: continue
        k = k.split("(")[0][-60:]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-34s n=%-3d avg %.4g" % (c, len(v), sum(v) / len(v)))
