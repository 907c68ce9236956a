#!/usr/bin/env python3
"""Upper bound for a merged conv dgrad + wgrad launch: the two launches of a block's pooling backward serial on one
stream against concurrent on two streams (they are independent: both read dpre, one writes dqkv, the other dw).
   python tools/pool_bwd_overlap.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svit_amd import ops
from tools.bench_kernels import rnd, B, DEV

def _streaming():
    """the streaming conv-backward wrappers (diagnostic build only since round 6: tools/diag/pool_streaming.py)"""
    from tools.diag import pool_streaming
    return pool_streaming


cfgs = {0: (1, (8, 56, 56), 1, 8), 1: (2, (8, 56, 56), 2, 4), 2: (2, (8, 28, 28), 1, 4), 3: (4, (8, 28, 28), 2, 2),
        4: (4, (8, 14, 14), 1, 2), 14: (8, (8, 14, 14), 2, 1)}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for blk, (h, thw, sq, skv) in cfgs.items():
    n_obj = 64
    N = 1 + thw[0] * thw[1] * thw[2] + n_obj
    qkv = rnd(B, N, 3, h, 96)
    ws = [torch.randn(96, 27, device=DEV) * 0.2 for _ in range(3)]
    strides = (sq, skv, skv)
    dpres = []
    for s in strides:
        no = 1 + thw[0] * ops.pooled(thw[1], s) * ops.pooled(thw[2], s) + n_obj
        dpres.append(rnd(B, h, no, 96))
    dws = [torch.zeros(96, 27, device=DEV) for _ in range(3)]
    dqkv = torch.zeros_like(qkv)
    wsp = torch.empty(8 << 20, device=DEV)
    def dg(): _streaming().pool_conv_dgrad_qkv(dpres, ws, dqkv, B, h, thw, n_obj, strides)
    def wg(): _streaming().pool_conv_wgrad_qkv(dpres, qkv, dws, B, h, thw, n_obj, strides, ws=wsp)
    def timed(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    def serial(): dg(); wg()
    def conc():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): dg()
        with torch.cuda.stream(s2): wg()
        cur.wait_stream(s1); cur.wait_stream(s2)
    def conc_batch(n=20):
        for _ in range(2): dg(); wg()
        torch.cuda.synchronize()
        cur = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            for _ in range(n): dg()
        with torch.cuda.stream(s2):
            for _ in range(n): wg()
        cur.wait_stream(s1); cur.wait_stream(s2)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    print("blk%-2d 20 dgrads beside 20 wgrads, one fork/join: %6.1f us per pair" % (blk, conc_batch()), flush=True)
    print("blk%-2d dgrad %6.1f us  wgrad %6.1f us  serial %6.1f us  two streams %6.1f us" %
          (blk, timed(dg), timed(wg), timed(serial), timed(conc)), flush=True)
