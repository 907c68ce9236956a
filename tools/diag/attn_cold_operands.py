#!/usr/bin/env python3
# NOTE (round 5): svit_attn_debug_set(2, m) -- "run only one of the two backward kernels" -- left the product library (it leaves outputs
# unwritten).  Build the two timing variants instead:  python tools/diag/build_variant.py dqonly attn_bwd.hip -DSVIT_DIAG_BWD_ONLY=1  (and =2 for dkv only)
# and run this script with SVIT_HIP_LIB pointing at them; the calls below then return SVIT_ERR_ARG and change nothing.
"""Attention kernels at the 14x14-stage shape with COLD operands (the caches are swept between launches, as inside the
training step where the saved activations come from HBM) against warm ones (isolated loop) and against cold ones that a
read-only touch kernel pulled into the Infinity Cache just before.  GPU box: python tools/diag/attn_cold_operands.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svit_amd import hip, ops
from tools.bench_kernels import rnd, KSC, BF16, DEV
lib = hip.load()
if lib.svit_attn_debug_set(2, 0) != 0:      # the run-one-backward-kernel switch left every library in round 5
    sys.exit("%s: the loaded library refuses svit_attn_debug_set(2, .): its 'dq only' / 'dkv only' timings would really be both kernels. "
             "Time the -DSVIT_DIAG_BWD_ONLY=1 / =2 variants (tools/diag/build_variant.py) with tools/bench_kernels.py attn instead." % sys.argv[0])
B, h, Nq, Nk, DA, J = 8, 4, 1633, 457, 128, 22
qa, ka, v = rnd(B, h, Nq, DA), (rnd(B, h, Nk, DA).float() * KSC).to(BF16), rnd(B, h, Nk, 96)
qa[..., 96 + J:] = 0; ka[..., 96 + J:] = 0
ctx, lse2 = ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=J)
dctx = rnd(B, Nq, h * 96)
junk = torch.empty(768 * 1024 * 1024 // 4, device=DEV)       # 768 MB: three times the Infinity Cache


def sweep():
    junk.add_(1.0)


def touch(ts):
    for t in ts:
        t.view(torch.int32).sum()      # a read of every byte (one reduction launch per tensor)


def timed(fn, pre=None, iters=12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = []
    for _ in range(iters):
        if pre:
            pre()
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        tot.append(e0.elapsed_time(e1) * 1e3)
    tot.sort()
    return tot[len(tot) // 2]


for name, fn, cold_ts in (
        ("fwd", lambda: ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=J), [qa, ka, v]),
        ("bwd (dq + dkv)", lambda: ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, 96 ** -0.5, bias_cols=J), [qa, ka, v, ctx, dctx])):
    warm = timed(fn)
    cold = timed(fn, pre=sweep)
    pref = timed(fn, pre=lambda: (sweep(), touch(cold_ts)))
    print("%-16s warm %.1f us | cold (caches swept) %.1f us | cold + operands touched first %.1f us   (single launches timed with events)"
          % (name, warm, cold, pref), flush=True)
for only, nm in ((2, "dq kernel"), (1, "dkv kernel")):
    lib.svit_attn_debug_set(2, only)
    fn = lambda: ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, 96 ** -0.5, bias_cols=J)
    print("%-16s warm %.1f us | cold %.1f us | cold + touched %.1f us" % (nm, timed(fn), timed(fn, pre=sweep),
          timed(fn, pre=lambda: (sweep(), touch([qa, ka, v, ctx, dctx])))), flush=True)
lib.svit_attn_debug_set(2, 0)
