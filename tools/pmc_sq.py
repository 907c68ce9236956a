#!/usr/bin/env python3
"""Per-kernel SQ counter averages of rocprofv3 --pmc passes, with the derived ratios the attention
work needs (matrix-pipe busy share, VALU / wait shares of the wave cycles).
usage: pmc_sq.py <csv> [<csv> ...] [--match substring]
Units (MI355X_MICROARCH.md, cycle-constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per 32x32x16 bf16 MFMA) summed
over SIMDs; SQ_BUSY_CYCLES is per-SE busy time summed over the 32 shader engines (x 8 XCD reporting)."""
import csv, sys, collections, re
match = None
files = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--match": match = args.pop(0)
    else: files.append(a)


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", n)[:70]


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if match and match not in k: continue
        agg[short(k)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(agg.items()):
    print(k)
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    for c, v in sorted(cs.items()):
        print("   %-34s n=%-3d avg %.5g" % (c, len(v), m[c]))
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
            if c in m: print("   -> %-28s / SQ_WAVE_CYCLES = %.3f" % (c, m[c] / wc))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CU_CYCLES" in m and m["SQ_BUSY_CU_CYCLES"]:
        print("   -> matrix pipe busy = MFMA_BUSY / (4 * BUSY_CU_CYCLES) = %.3f" %
              (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * m["SQ_BUSY_CU_CYCLES"])))
    if "SQ_INSTS_MFMA" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m and m["SQ_INSTS_MFMA"]:
        print("   -> MFMA_BUSY per MFMA instruction = %.2f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_INSTS_MFMA"]))
