#!/usr/bin/env python3
"""Forward attention time at the 14x14-stage shape (B 8, h 4, Nq 1633, DA 128) for Nk = 64 (one tile: the fixed
cost) and Nk = 457 (the real launch).  Run once per library variant:
  SVIT_HIP_LIB=tools/diag/libsvit_diag_attnabl<mask>.so python tools/diag/attn_fwd_ablate.py <label>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import warnings
warnings.filterwarnings("ignore")
import torch
from svit_amd import ops
from tools.bench_kernels import rnd, timeit, KSC, BF16
label = sys.argv[1] if len(sys.argv) > 1 else "product"
out = []
for (B, h, Nq, DA, J) in [(8, 4, 1633, 128, 22), (8, 2, 6337, 160, 36)]:
    for Nk in (64, 457, 1633):
        qa, ka, v = rnd(B, h, Nq, DA), (rnd(B, h, Nk, DA).float() * KSC).to(BF16), rnd(B, h, Nk, 96)
        f = min(timeit(lambda: ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=J), iters=40) for _ in range(3))
        out.append("Nq%d/Nk%d %.1f" % (Nq, Nk, f))
print("%-22s " % label + "  ".join(out), flush=True)
