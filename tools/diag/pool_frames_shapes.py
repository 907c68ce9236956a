#!/usr/bin/env python3
"""The pooling forward launches of the frames pass (B*T = 128 single frames, T' = 1, save=False) one by one: the fused q/k/v
launch, each tensor alone, with / without the rel-pos columns -- where do the ~98 us per block go?  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svit_amd import hip, ops
from tools.bench_kernels import rnd, timeit, DEV
Bf, n_obj = 128, 4
for blk, h, thw, sq, skv in [(0, 1, (1, 56, 56), 1, 8), (1, 2, (1, 56, 56), 2, 4), (2, 2, (1, 28, 28), 1, 4), (3, 4, (1, 28, 28), 2, 2),
                             (4, 4, (1, 14, 14), 1, 2), (14, 8, (1, 14, 14), 2, 1), (15, 8, (1, 7, 7), 1, 1)]:
    N = 1 + thw[0] * thw[1] * thw[2] + n_obj
    qkv = rnd(Bf, N, 3, h, 96)
    ws = [torch.randn(96, 27, device=DEV) * 0.2 for _ in range(3)]
    g = [torch.ones(96, device=DEV) for _ in range(3)]
    b = [torch.zeros(96, device=DEV) for _ in range(3)]
    J = 2 * ops.pooled(thw[1], skv) + thw[0]
    da = 128 if J <= 32 else 160
    strides, lds, modes = (sq, skv, skv), (da, da, 96), (0, 1, 0)
    t1 = [timeit(lambda i=i: ops.pool_ln_fwd(qkv, i, ws[i], g[i], b[i], Bf, h, thw, n_obj, strides[i], ld_out=lds[i], mode=modes[i])) for i in range(3)]
    tf = timeit(lambda: ops.pool_ln_fwd_qkv(qkv, ws, g, b, Bf, h, thw, n_obj, strides, lds, modes, save=False))
    ts = timeit(lambda: ops.pool_ln_fwd_qkv(qkv, ws, g, b, Bf, h, thw, n_obj, strides, lds, modes, save=True))
    Nq = 1 + ops.pooled(thw[1], sq) * ops.pooled(thw[2], sq) + n_obj
    Nk = 1 + ops.pooled(thw[1], skv) * ops.pooled(thw[2], skv) + n_obj
    mb = (qkv.numel() * 2 + Bf * h * (Nq * da + Nk * da + Nk * 96) * 2) / 1e6
    print("blk%-2d h=%d N=%5d Nq=%5d Nk=%3d | q %6.1f k %6.1f v %6.1f | fused no-save %6.1f  save %6.1f us | %.0f MB -> %.2f TB/s" %
          (blk, h, N, Nq, Nk, t1[0], t1[1], t1[2], tf, ts, mb, mb / tf), flush=True)      # MB / us = TB/s
