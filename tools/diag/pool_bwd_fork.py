#!/usr/bin/env python3
"""pool conv dgrad3 and wgrad3 (independent: both read dpre) one after the other against side by side on two
streams, at the 14x14-stage shape: is there anything for ONE launch holding both kinds of workgroup to win?  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svit_amd import hip, ops
from tools.bench_kernels import rnd, timeit, DEV
B, n_obj = 8, 64

def _streaming():
    """the streaming conv-backward wrappers (diagnostic build only since round 6: tools/diag/pool_streaming.py)"""
    from tools.diag import pool_streaming
    return pool_streaming


side = torch.cuda.Stream()
for blk, h, thw, sq, skv in [(4, 4, (8, 14, 14), 1, 2), (2, 2, (8, 28, 28), 1, 4), (14, 8, (8, 14, 14), 2, 1)]:
    N = 1 + thw[0] * thw[1] * thw[2] + n_obj
    qkv = rnd(B, N, 3, h, 96)
    ws = [torch.randn(96, 27, device=DEV) * 0.2 for _ in range(3)]
    strides = (sq, skv, skv)
    dpres = [rnd(B, h, 1 + thw[0] * ops.pooled(thw[1], s) * ops.pooled(thw[2], s) + n_obj, 96) for s in strides]
    dqkv = torch.empty_like(qkv)
    dws = [torch.zeros(96, 27, device=DEV) for _ in range(3)]
    wsp = torch.empty(8 << 20, device=DEV)
    dg = lambda: _streaming().pool_conv_dgrad_qkv(dpres, ws, dqkv, B, h, thw, n_obj, strides)
    wg = lambda: _streaming().pool_conv_wgrad_qkv(dpres, qkv, dws, B, h, thw, n_obj, strides, wsp)
    def serial():
        dg(); wg()
    def forked():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            wg()
        dg()
        main.wait_stream(side)
    def fork_only():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        main.wait_stream(side)
    r = {n: min(timeit(f, iters=30) for _ in range(3)) for n, f in (("dgrad3", dg), ("wgrad3", wg), ("serial", serial), ("forked", forked), ("empty fork+join", fork_only))}
    print("blk%-2d h=%d N=%5d sq=%d skv=%d | " % (blk, h, N, sq, skv) + "  ".join("%s %.1f us" % kv for kv in r.items()), flush=True)
