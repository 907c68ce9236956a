#!/usr/bin/env python3
"""slab pooling forward vs the streaming kernels (svit_debug_set_pool) on a list of shapes: max |diff| of
out / pre / mean / rstd per tensor.  python tools/pool_slab_check.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svit_amd import ops, hip
lib = hip.load()
lib.svit_debug_set_pool.restype, lib.svit_debug_set_pool.argtypes = C.c_int32, [C.c_int32, C.c_int32]
DEV = "cuda"
torch.manual_seed(0)
for (B, h, thw, sq, skv, n_obj) in [(2, 4, (8, 14, 14), 1, 2, 64), (1, 4, (16, 14, 14), 1, 2, 128), (2, 8, (16, 14, 14), 2, 1, 128),
                                    (3, 4, (1, 14, 14), 1, 2, 4), (2, 8, (8, 7, 7), 1, 1, 64), (1, 8, (16, 7, 7), 1, 1, 128),
                                    (2, 2, (2, 8, 8), 1, 2, 8), (2, 1, (3, 10, 10), 2, 1, 12), (1, 4, (8, 13, 9), 1, 2, 8)]:
    N = 1 + thw[0] * thw[1] * thw[2] + n_obj
    qkv = (torch.randn(B, N, 3, h, 96, device=DEV) * 0.5).bfloat16()
    ws = [torch.randn(96, 27, device=DEV) * 0.2 for _ in range(3)]
    g = [torch.rand(96, device=DEV) + 0.5 for _ in range(3)]
    b = [torch.randn(96, device=DEV) * 0.1 for _ in range(3)]
    wflat = torch.cat([w.flatten() for w in ws]).contiguous()
    offs = torch.tensor([0, 2592, 5184], dtype=torch.int64, device=DEV)
    sel = ops.pool_weight_sel(wflat, offs, torch.zeros((3, 2592), dtype=torch.int32, device=DEV))
    sels = [sel[i] for i in range(3)]
    J = 2 * ops.pooled(thw[1], skv) + thw[0]
    da = 128 if J <= 32 else 160
    res = []
    for on in (0, 1):
        lib.svit_debug_set_pool(0, on)
        r = ops.pool_ln_fwd_qkv(qkv, ws, g, b, B, h, thw, n_obj, (sq, skv, skv), (da, da, 96), (0, 1, 0), sels=sels, out_scales=(1.0, 0.147, 1.0))
        torch.cuda.synchronize()
        res.append([[t.float().clone() for t in tup] for tup in r])
    lib.svit_debug_set_pool(0, 1)
    msg = []
    for i in range(3):
        d = [float((res[0][i][k] - res[1][i][k]).abs().nan_to_num(1e9).max()) for k in range(4)]
        # out's bias columns of q are uninitialised in both paths (the gather fills them): compare 0..95 (+one-hot for k)
        cols = slice(0, 96) if i == 0 else slice(None)
        d[0] = float((res[0][i][0][..., cols] - res[1][i][0][..., cols]).abs().nan_to_num(1e9).max())
        msg.append("%s out %.1e pre %.1e mean %.1e rstd %.1e" % ("qkv"[i], d[0], d[1], d[2], d[3]))
    print("B%d h%d thw %s sq%d skv%d: " % (B, h, thw, sq, skv) + " | ".join(msg), flush=True)
