#!/usr/bin/env python3
"""Per-kernel mean of a rocprofv3 --pmc counter (…_counter_collection.csv).
    python tools/pmc_summary.py <csv> [<csv> ...]
Prints kernel, counter, launches, mean per launch, total.  FETCH_SIZE / WRITE_SIZE are reported
by rocprofv3 in KiB-like units of 1 KB; MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE
counts half of the bytes of wide streaming reads -> the 'corrected' column doubles it."""
import csv
import sys
from collections import defaultdict


def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]


def main():
    for path in sys.argv[1:]:
        agg = defaultdict(lambda: [0.0, 0])
        with open(path) as f:
            for r in csv.DictReader(f):
                key = (short(r["Kernel_Name"]), r["Counter_Name"])
                agg[key][0] += float(r["Counter_Value"])
                agg[key][1] += 1
        print("# " + path)
        print("%-62s %-12s %8s %14s %14s %14s" % ("kernel", "counter", "launches", "mean/launch", "corrected", "total"))
        for (k, c), (tot, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
            mean = tot / n
            corr = mean * 2 if c == "FETCH_SIZE" else mean
            print("%-62s %-12s %8d %14.1f %14.1f %14.1f" % (k, c, n, mean, corr, tot))


if __name__ == "__main__":
    main()
