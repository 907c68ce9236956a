#!/usr/bin/env python3
"""Runs the fused attention forward + backward at every block shape of a config a few times
(for tools/attn_trace.sh).  python tools/attn_shapes.py [c2|c4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svit_amd import ops
from tools.bench_kernels import BLOCKS, rnd, B
C4 = [(0, 50305, 50305, 913, 96, 96, 1, 128), (1, 50305, 12673, 3265, 96, 192, 2, 160),
      (2, 12673, 12673, 913, 192, 192, 2, 128), (3, 12673, 3265, 3265, 192, 384, 4, 160),
      (4, 3265, 3265, 913, 384, 384, 4, 128), (14, 3265, 913, 3265, 384, 768, 8, 160),
      (15, 913, 913, 913, 768, 768, 8, 128)]
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
blocks, b = (C4, 4) if which == "c4" else (BLOCKS, B)
for blk, Nin, Nq, Nk, Ci, Co, h, DA in blocks:
    J = 22 if DA == 128 else 36
    if which == "c4":
        J = 32 + 7 + 7 - 0 if DA == 128 else 32 + 14 + 14
        J = min(J, DA - 96)
    qa, ka, v = rnd(b, h, Nq, DA), (rnd(b, h, Nk, DA).float() * 0.1472).to(torch.bfloat16), rnd(b, h, Nk, 96)
    qa[..., 96 + J:] = 0
    ka[..., 96 + J:] = 0
    dctx = rnd(b, Nq, h * 96)
    for _ in range(5):
        ctx, lse2 = ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=J)
        ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, 96 ** -0.5, bias_cols=J)
    torch.cuda.synchronize()
