#!/usr/bin/env python3
"""One block shape of the q/k/v pooling forward (slab path) and of the conv backward (dgrad + wgrad) a few times,
for rocprofv3 --kernel-trace.   python tools/pool_one.py [blk]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svit_amd import ops
from tools.bench_kernels import rnd, B, DEV
cfgs = {0: (1, (8, 56, 56), 1, 8), 1: (2, (8, 56, 56), 2, 4), 2: (2, (8, 28, 28), 1, 4), 3: (4, (8, 28, 28), 2, 2),
        4: (4, (8, 14, 14), 1, 2), 14: (8, (8, 14, 14), 2, 1), 15: (8, (8, 7, 7), 1, 1)}
blk = int(sys.argv[1]) if len(sys.argv) > 1 else 4
h, thw, sq, skv = cfgs[blk]
n_obj = 64
N = 1 + thw[0] * thw[1] * thw[2] + n_obj
qkv = rnd(B, N, 3, h, 96)
ws = [torch.randn(96, 27, device=DEV) * 0.2 for _ in range(3)]
g = [torch.ones(96, device=DEV) for _ in range(3)]
b = [torch.zeros(96, device=DEV) for _ in range(3)]
wflat = torch.cat([w.flatten() for w in ws]).contiguous()
offs = torch.tensor([0, 2592, 5184], dtype=torch.int64, device=DEV)
sel = ops.pool_weight_sel(wflat, offs, torch.zeros((3, 2592), dtype=torch.int32, device=DEV))
sels = [sel[i] for i in range(3)]
J = 2 * ops.pooled(thw[1], skv) + thw[0]
da = 128 if J <= 32 else 160
for _ in range(6):
    outs = ops.pool_ln_fwd_qkv(qkv, ws, g, b, B, h, thw, n_obj, (sq, skv, skv), (da, da, 96), (0, 1, 0), sels=sels)
dpres = [torch.randn_like(outs[i][1].float()).to(outs[i][1].dtype) for i in range(3)]
dws = [torch.zeros(96, 27, device=DEV) for _ in range(3)]
dqkv = torch.zeros_like(qkv)
for _ in range(6):
    ops.pool_conv_bwd_qkv(dpres, ws, dqkv, qkv, dws, B, h, thw, n_obj, (sq, skv, skv))
torch.cuda.synchronize()
