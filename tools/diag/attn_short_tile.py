#!/usr/bin/env python3
"""The frames pass's attention forward launches (Nk = 54): the T' = 1 tile (svit_attn_debug_set(3, 1)) against the generic
128-query kernel (3, 0), isolated, us.  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svit_amd import hip, ops
from tools.bench_kernels import rnd, timeit, KSC, BF16, DEV
lib = hip.load()
for (B, h, Nq, Nk) in [(128, 1, 3141, 54), (128, 2, 789, 54), (128, 4, 201, 54), (128, 8, 54, 54), (63, 4, 201, 54)]:
    qa, ka, v = rnd(B, h, Nq, 128), (rnd(B, h, Nk, 128).float() * KSC).to(BF16), rnd(B, h, Nk, 96)
    qa[..., 111:] = 0; ka[..., 111:] = 0
    res = []
    for on in (0, 1):
        lib.svit_attn_debug_set(3, on)
        res.append(min(timeit(lambda: ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=15), iters=30) for _ in range(3)))
    lib.svit_attn_debug_set(3, 1)
    mb = (qa.numel() * 2 * 1.75 + B * Nq * h * 96 * 2 + (ka.numel() + v.numel()) * 2) / 1e6    # Q read + its 96 residual columns again + ctx + K/V
    print("B %3d h %d Nq %4d Nk %d | generic %6.1f us  T'=1 tile %6.1f us | %.0f MB -> %.2f TB/s" % (B, h, Nq, Nk, res[0], res[1], mb, mb / res[1]), flush=True)      # MB / us = TB/s
