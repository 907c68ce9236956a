#!/usr/bin/env python3
"""Per-workgroup wall-clock anatomy of pool_bwd_fused_kernel (s_memrealtime stamps, diagnostic build):
    python tools/diag/build_variant.py poolstamps pool.hip -DSVIT_POOL_STAMPS      (here)
    python tools/pool_bwd_stamps.py [blk]                                            (GPU box)
prints start / stage / walk / tail times of the workgroups by tensor."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_path = os.path.join(ROOT, "tools", "diag", "libsvit_diag_poolstamps.so")
os.environ["SVIT_HIP_LIB"] = lib_path
from svit_amd import hip, ops

DEV = torch.device("cuda")
lib = hip.load()
SH = {"blk0": (1, (8, 56, 56), 1, 8), "blk1": (2, (8, 56, 56), 2, 4), "blk2": (2, (8, 28, 28), 1, 4), "blk3": (4, (8, 28, 28), 2, 2), "blk4": (4, (8, 14, 14), 1, 2), "blk14": (8, (8, 14, 14), 2, 1), "blk15": (8, (8, 7, 7), 1, 1)}
name = sys.argv[1] if len(sys.argv) > 1 else "blk4"
h, thw, sq, skv = SH[name]
B, n_obj = 8, 64
N = 1 + thw[0] * thw[1] * thw[2] + n_obj
g = torch.Generator(device="cpu").manual_seed(1)
qkv = (torch.randn((B, N, 3, h, 96), generator=g) * 0.5).to(DEV, torch.bfloat16)
ws = [(torch.randn((96, 27), generator=g) * 0.2).to(DEV) for _ in range(3)]
strides = (sq, skv, skv)
dpres = [torch.randn((B, h, 1 + thw[0] * ops.pooled(thw[1], s) * ops.pooled(thw[2], s) + n_obj, 96), generator=g).to(DEV, torch.bfloat16)
         for s in strides]
dqkv = torch.empty_like(qkv)
dws = [torch.zeros((96, 27), device=DEV) for _ in range(3)]
wsb = torch.empty(9 * 1024 * 1024, device=DEV)
for _ in range(3):
    ops.pool_conv_bwd_qkv(dpres, ws, dqkv, qkv, dws, B, h, thw, n_obj, strides, ws=wsb)
torch.cuda.synchronize()
n = 8 * 2048
buf = (ctypes.c_ulonglong * n)()
raw = ctypes.CDLL(lib_path)
assert raw.svit_debug_pool_bwd_wg_times(buf, n) == 0
w = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(2048, 8)
w = w[w[:, 3] > 0]
t0 = w[:, 0].min()
us = lambda a: (a - t0) / 100.0
print("%s: %d workgroups, kernel span %.1f us" % (name, len(w), us(w[:, 3]).max()))
for which in range(3):
    m = w[w[:, 4] == which]
    if not len(m):
        continue
    st, sg, wk, en = us(m[:, 0]), us(m[:, 1]), us(m[:, 2]), us(m[:, 3])
    print("  tensor %d (stride %d): %4d WGs  start med %.1f max %.1f | stage med %.1f max %.1f | walk med %.1f max %.1f | tail med %.1f | total med %.1f max %.1f | end max %.1f"
          % (which, strides[which], len(m), np.median(st), st.max(), np.median(sg - st), (sg - st).max(), np.median(wk - sg), (wk - sg).max(),
             np.median(en - wk), np.median(en - st), (en - st).max(), en.max()))
hw = w[:, 6]
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print("  distinct (se, sh, cu) triples: %d" % len(set(zip(se.tolist(), sh.tolist(), cu.tolist()))))
