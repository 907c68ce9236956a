#!/usr/bin/env python3
"""Attention forward / backward time against the number of key tiles at fixed query shape:
intercept = per-launch fixed cost, slope = cost per 64-key tile.  python tools/attn_vs_nk.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svit_amd import ops
from tools.bench_kernels import rnd, timeit, KSC, BF16
for (B, h, Nq, DA, J) in [(8, 4, 1633, 128, 22), (8, 2, 6337, 160, 36), (8, 1, 25153, 128, 22)]:
    print("B=%d h=%d Nq=%d DA=%d" % (B, h, Nq, DA))
    for Nk in (64, 128, 256, 448, 512, 1024, 1600):
        qa, ka, v = rnd(B, h, Nq, DA), (rnd(B, h, Nk, DA).float() * KSC).to(BF16), rnd(B, h, Nk, 96)
        qa[..., 96 + J:] = 0
        ka[..., 96 + J:] = 0
        f = timeit(lambda: ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=J), iters=30)
        ctx, lse2 = ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=J)
        dctx = rnd(B, Nq, h * 96)
        b = timeit(lambda: ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, 96 ** -0.5, bias_cols=J), iters=15)
        print("  Nk=%5d tiles=%3d  fwd %7.1f us (%.2f us/tile)  bwd %7.1f us" % (Nk, Nk // 64, f, f / (Nk // 64), b), flush=True)
