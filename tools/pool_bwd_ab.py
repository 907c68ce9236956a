#!/usr/bin/env python3
"""Pooling conv backward of q, k, v at the step's shapes: the fused kernel (pool_bwd_fused_kernel, one launch) against the
streaming launches it replaces (pool_dgrad3 + pool_wgrad3; svit_debug_set_pool(1, 0)), isolated loops, us per call
(incl. the second-stage reduce).  GPU box.   python tools/pool_bwd_ab.py [B]"""
# NOTE (round 6): the knob-off arm (svit_debug_set_pool(1, 0) -> the streaming launches) exists only in a -DSVIT_DIAG_POOL_STREAMING build
# (python tools/diag/build_variant.py poolstream pool.hip -DSVIT_DIAG_POOL_STREAMING; SVIT_HIP_LIB=tools/diag/libsvit_diag_poolstream.so):
# against the product library the entry point returns SVIT_ERR_SHAPE with the fused kernel refused, and this tool stops there.
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svit_amd import hip, ops

DEV = torch.device("cuda")
BF16 = torch.bfloat16
lib = hip.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [("blk0", 1, (8, 56, 56), 1, 8), ("blk1", 2, (8, 56, 56), 2, 4), ("blk2", 2, (8, 28, 28), 1, 4),
          ("blk3", 4, (8, 28, 28), 2, 2), ("blk4-13", 4, (8, 14, 14), 1, 2), ("blk14", 8, (8, 14, 14), 2, 1),
          ("blk15", 8, (8, 7, 7), 1, 1), ("c4 blk4-13", 4, (16, 14, 14), 1, 2), ("frames blk4-13 (B*16)", 4, (1, 14, 14), 1, 2)]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, h, thw, sq, skv in SHAPES:
    b = B * 16 if name.startswith("frames") else (B // 2 if name.startswith("c4") else B)
    n_obj = 4 if thw[0] == 1 else thw[0] * 2 * 4
    L = thw[0] * thw[1] * thw[2]
    N = 1 + L + n_obj
    g = torch.Generator(device="cpu").manual_seed(1)
    qkv = (torch.randn((b, N, 3, h, 96), generator=g) * 0.5).to(DEV, BF16)
    ws = [(torch.randn((96, 27), generator=g) * 0.2).to(DEV) for _ in range(3)]
    strides = (sq, skv, skv)
    dpres = []
    for s in strides:
        nout = 1 + thw[0] * ops.pooled(thw[1], s) * ops.pooled(thw[2], s) + n_obj
        dpres.append(torch.randn((b, h, nout, 96), generator=g).to(DEV, BF16))
    dqkv = torch.empty_like(qkv)
    dws = [torch.zeros((96, 27), device=DEV) for _ in range(3)]
    ws_buf = torch.empty(18 * 1024 * 1024, device=DEV)
    run = lambda: ops.pool_conv_bwd_qkv(dpres, ws, dqkv, qkv, dws, b, h, thw, n_obj, strides, ws=ws_buf)
    try:
        lib.svit_debug_set_pool(1, 1)
        t_f = timeit(run)
        lib.svit_debug_set_pool(1, 0)
        t_s = timeit(run)
    finally:
        lib.svit_debug_reset()
    mb = (qkv.numel() * 2 * 2 + sum(d.numel() for d in dpres) * 2) / 1e6
    print("%-24s B=%-3d h=%d thw=%-12s s=(%d,%d)  fused %7.1f us   streaming %7.1f us   (%.0f MB algorithmic -> %.2f TB/s fused)"
          % (name, b, h, thw, sq, skv, t_f, t_s, mb, mb / t_f))
