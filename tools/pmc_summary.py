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


FAMILIES = (("gemm_nt", "NT GEMM"), ("gemm_tn", "TN GEMM"), ("attn_", "attention"), ("relq", "rel-pos plumbing"),
            ("pool_", "pooling"), ("ln_", "LayerNorm"), ("adamw", "optimizer"), ("sumsq", "optimizer"),
            ("reduce_partials", "reductions"))


def family(name):
    for k, v in FAMILIES:
        if k in name:
            return v
    return "other"


def family_table(paths, steps):
    """GB per step and kernel family: FETCH_SIZE x 2 (the guide's gfx950 correction for wide streaming reads; an upper bound
    for gather reads) + WRITE_SIZE, the counter unit being 1 KB.  Wasted re-reads show here before they show in time."""
    fam = defaultdict(lambda: [0.0, 0.0])
    for path in paths:
        with open(path) as f:
            for r in csv.DictReader(f):
                c = r["Counter_Name"]
                if c in ("FETCH_SIZE", "WRITE_SIZE"):
                    fam[family(short(r["Kernel_Name"]))][0 if c == "FETCH_SIZE" else 1] += float(r["Counter_Value"])
    print("# counter traffic per step and kernel family (GB; fetch = FETCH_SIZE x 2), %d step(s) traced" % steps)
    print("%-18s %10s %10s %10s" % ("family", "fetch", "write", "total"))
    tot = [0.0, 0.0]
    for k, (fe, wr) in sorted(fam.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1])):
        fe, wr = 2 * fe * 1e3 / 1e9 / steps, wr * 1e3 / 1e9 / steps
        tot[0] += fe
        tot[1] += wr
        print("%-18s %10.2f %10.2f %10.2f" % (k, fe, wr, fe + wr))
    print("%-18s %10.2f %10.2f %10.2f" % ("all", tot[0], tot[1], tot[0] + tot[1]))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--steps=")]
    steps = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--steps=")), 2)   # --steps 1 --warmup 1 = two eager steps
    family_table(args, steps)
    for path in args:
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
