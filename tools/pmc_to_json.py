#!/usr/bin/env python3
"""profiles/<name>_pmc_traffic.txt (tools/pmc_summary.py output) -> profiles/pmc_traffic.json:
mean corrected bytes per launch for the C-ABI entries bench.py reports a roofline for."""
import json
import re
import sys

src = sys.argv[1]
git_head = sys.argv[2] if len(sys.argv) > 2 else None
fam_of = lambda k: ("svit_gemm_nt" if "gemm_nt_v2" in k or "gemm_nt_ring" in k else
                    "svit_attn_fwd" if "attn_fwd_kernel" in k or "attn_fwd2_kernel" in k else
                    "svit_attn_bwd" if "attn_bwd_" in k else
                    "svit_gemm_tn_grouped" if "gemm_tn_grouped" in k else
                    "svit_pool_ln_fwd_qkv" if any(t in k for t in ("pool_ln_fwd3", "pool_fwd_staged", "pool_slab_", "pool_mfma_fwd",
                                                                    "pool_frame_fwd")) else
                    "svit_pool_conv_bwd_qkv" if "pool_dgrad3" in k or "pool_wgrad3" in k or "pool_bwd_fused" in k else None)
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
       "git_head": git_head, "bytes_per_launch": {}}
for fam, a in agg.items():
    out["bytes_per_launch"][fam] = {"fetch": round(a["FETCH_SIZE"][0] / max(1, a["FETCH_SIZE"][1])),
                                    "write": round(a["WRITE_SIZE"][0] / max(1, a["WRITE_SIZE"][1])),
                                    "launches_sampled": a["FETCH_SIZE"][1]}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["bytes_per_launch"], indent=1))
