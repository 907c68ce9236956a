#!/usr/bin/env python3
"""profiles/<name>_pmc_traffic.txt (tools/pmc_summary.py output) -> profiles/pmc_traffic.json:
mean corrected bytes per launch for the C-ABI entries bench.py reports a roofline for."""
import json
import re
import sys

src = sys.argv[1]
fam_of = lambda k: ("svit_gemm_nt" if k.startswith("gemm_nt_v2") else
                    "svit_attn_fwd" if k.startswith("attn_fwd_kernel") else
                    "svit_gemm_tn_grouped" if k.startswith("gemm_tn_grouped") else None)
agg = {}
for line in open(src):
    m = re.match(r"(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", line)
    if not m:
        continue
    k, c, n, mean, corr, tot = m.groups()
    fam = fam_of(k)
    if not fam:
        continue
    a = agg.setdefault(fam, {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
    a[c][0] += float(tot) * 1e3 * (2 if c == "FETCH_SIZE" else 1)   # KB -> bytes, gfx950 FETCH x2
    a[c][1] += int(n)
out = {"source": "%s: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of "
                 "`bench.py --steps 1 --warmup 1 --eager`; KB units, FETCH_SIZE doubled "
                 "(MI355X_MICROARCH.md, HBM section: gfx950 counts 64 B per 128-B read request)" % src,
       "bytes_per_launch": {}}
for fam, a in agg.items():
    out["bytes_per_launch"][fam] = {"fetch": round(a["FETCH_SIZE"][0] / max(1, a["FETCH_SIZE"][1])),
                                    "write": round(a["WRITE_SIZE"][0] / max(1, a["WRITE_SIZE"][1])),
                                    "launches_sampled": a["FETCH_SIZE"][1]}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["bytes_per_launch"], indent=1))
