#!/usr/bin/env python3
"""Run the fused attention forward (and optionally backward) a few times at one block shape
(for rocprofv3 --pmc runs).  usage: attn_one.py <blk> [bwd]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from svit_amd import hip, ops
from tools.bench_kernels import BLOCKS, B, rnd
blk = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = [c for c in BLOCKS if c[0] == blk][0]
_, Nin, Nq, Nk, Ci, Co, h, DA = cfg
qa, ka, v = rnd(B, h, Nq, DA), rnd(B, h, Nk, DA), rnd(B, h, Nk, 96)
scale = 96 ** -0.5
for _ in range(5):
    ctx, lse2 = ops.attn_fwd(qa, ka, v, scale)
if len(sys.argv) > 2:
    dctx = rnd(B, Nq, h * 96)
    for _ in range(3):
        ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, scale)
torch.cuda.synchronize()
