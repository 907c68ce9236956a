#!/usr/bin/env python3
"""Run the fused attention forward (and optionally backward) a few times at one block shape
(for rocprofv3 --pmc runs).  usage: attn_one.py <blk> [bwd] [ap]     (ap: force the anti-phase forward of a library built from
tools/diag/variants/attn_fwd_anti_phase.patch -- svit_attn_debug_set(4, 1), (5, 0))"""
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
if "ap" in sys.argv[2:]:
    lib = hip.load()
    assert lib.svit_attn_debug_set(4, 1) == 0 and lib.svit_attn_debug_set(5, 0) == 0, "this library has no anti-phase forward"
for _ in range(5):
    ctx, lse2 = ops.attn_fwd(qa, ka, v, scale)
if "bwd" in sys.argv[2:]:
    dctx = rnd(B, Nq, h * 96)
    for _ in range(3):
        ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, scale)
torch.cuda.synchronize()
