#!/usr/bin/env python3
"""dkv-kernel time against the number of query stages at a fixed key shape (rocprofv3 kernel trace
is the caller's job; here: whole backward minus nothing, us).  Slope = cost per 64-query stage."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svit_amd import ops, hip
from tools.bench_kernels import rnd, timeit, KSC, BF16
lib = hip.load()
lib.svit_attn_debug_set.restype, lib.svit_attn_debug_set.argtypes = C.c_int32, [C.c_int32, C.c_int32]
B, h, DA, J, Nk = 8, 4, 128, 22, 457
for halves in (1, 2):
    lib.svit_attn_debug_set(0, halves)
    for Nq in (64, 128, 256, 512, 1024, 1633, 3266):
        qa, ka, v = rnd(B, h, Nq, DA), (rnd(B, h, Nk, DA).float() * KSC).to(BF16), rnd(B, h, Nk, 96)
        ctx, lse2 = ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=J)
        dctx = rnd(B, Nq, h * 96)
        t = timeit(lambda: ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, 96 ** -0.5, q_splits=1, bias_cols=J), iters=15)
        print("halves %d Nq=%5d stages=%3d  bwd %7.1f us" % (halves, Nq, (Nq + 63) // 64, t), flush=True)
lib.svit_attn_debug_set(0, 0)
