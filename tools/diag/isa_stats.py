#!/usr/bin/env python3
"""Per-kernel register / spill / LDS figures from a hipcc -S file (tools/diag/isa_of.sh): python tools/diag/isa_stats.py /tmp/isa/x.s [filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
    blk = m.group(0)
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if flt in name:
        print("%-90s vgpr %s agpr %s sgpr %s spill v%s s%s scratch %s" % (name[:90], g("vgpr_count"), g("agpr_count"), g("sgpr_count"),
              g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")))
