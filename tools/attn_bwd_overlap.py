#!/usr/bin/env python3
# NOTE (round 5): svit_attn_debug_set(2, m) -- "run only one of the two backward kernels" -- left the product library (it leaves outputs
# unwritten).  Build the two timing variants instead:  python tools/diag/build_variant.py dqonly attn_bwd.hip -DSVIT_DIAG_BWD_ONLY=1  (and =2 for dkv only)
# and run this script with SVIT_HIP_LIB pointing at them; the calls below then return SVIT_ERR_ARG and change nothing.
"""Upper bound for running the dq and the dkv kernel of an attention backward side by side (today: two dependent launches,
the dq kernel writes the delta rows the dkv kernel reads): 20 dq launches on one stream beside 20 dkv launches on another
against the serial pair, per block shape.  (Timing only: the dkv launches read the delta rows of an earlier iteration.)
   python tools/attn_bwd_overlap.py [c2|c4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

def _need_bwd_only_switch(lib):
    """svit_attn_debug_set(2, m) ("run only one of the two backward kernels") left every library in round 5 -- the
    product refuses it, the diagnostic builds select the kernel at COMPILE time (-DSVIT_DIAG_BWD_ONLY=1 dq / 2 dkv).  Without the
    switch this tool would print "dq only" / "dkv only" timings that are really both kernels: refuse to run instead."""
    import sys
    if lib.svit_attn_debug_set(2, 0) != 0:
        sys.exit("%s: the loaded library has no run-one-backward-kernel switch (svit_attn_debug_set(2, .) -> SVIT_ERR_ARG); "
                 "time the -DSVIT_DIAG_BWD_ONLY=1 / =2 variants of tools/diag/build_variant.py with tools/bench_kernels.py attn "
                 "instead.  Not producing numbers." % sys.argv[0])

import torch
from svit_amd import ops, hip
from tools.bench_kernels import rnd, BLOCKS, BLOCKS_C4, KSC, BF16
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
blocks, B = (BLOCKS_C4, 4) if cfg == "c4" else (BLOCKS, 8)
lib = hip.load()
lib.svit_attn_debug_set.restype, lib.svit_attn_debug_set.argtypes = C.c_int32, [C.c_int32, C.c_int32]
_need_bwd_only_switch(lib)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
tot = [0.0, 0.0]
for blk, Nin, Nq, Nk, Ci, Co, h, DA in blocks:
    J = (30 if DA == 128 else 44) if cfg == "c4" else (22 if DA == 128 else 36)
    qa, ka, v = rnd(B, h, Nq, DA), (rnd(B, h, Nk, DA).float() * KSC).to(BF16), rnd(B, h, Nk, 96)
    qa[..., 96 + J:] = 0
    ka[..., 96 + J:] = 0
    scale = 96 ** -0.5
    ctx, lse2 = ops.attn_fwd(qa, ka, v, scale, bias_cols=J)
    dctx = rnd(B, Nq, h * 96)
    def bwd(): ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, scale, bias_cols=J)
    def timed(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    def conc(n=20):
        cur = torch.cuda.current_stream()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s1.wait_stream(cur); s2.wait_stream(cur)
        lib.svit_attn_debug_set(2, 2)
        with torch.cuda.stream(s1):
            for _ in range(n): bwd()
        lib.svit_attn_debug_set(2, 1)
        with torch.cuda.stream(s2):
            for _ in range(n): bwd()
        lib.svit_attn_debug_set(2, 0)
        cur.wait_stream(s1); cur.wait_stream(s2)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    ser = timed(bwd)
    lib.svit_attn_debug_set(2, 2); dq = timed(bwd)
    lib.svit_attn_debug_set(2, 1); dkv = timed(bwd)
    lib.svit_attn_debug_set(2, 0)
    conc(); c = conc()
    mult = 10 if blk == 4 else 1
    tot[0] += ser * mult; tot[1] += c * mult
    print("blk%-2d h=%d Nq=%6d Nk=%5d  dq %6.1f  dkv %6.1f  serial pair %6.1f  side by side %6.1f us" % (blk, h, Nq, Nk, dq, dkv, ser, c), flush=True)
print("step totals (16 launches): serial %.1f us, side by side %.1f us" % tuple(tot))
