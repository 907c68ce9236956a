#!/usr/bin/env python3
"""Per-item wall-clock anatomy of pool_frame_fwd_kernel (s_memrealtime stamps, diagnostic build):
    python tools/diag/build_variant.py poolstamps pool.hip -DSVIT_POOL_STAMPS      (here)
    python tools/pool_frame_stamps.py [blk]                                          (GPU box)
prints, for workgroup 0 and as medians over the workgroups: wait (top -> landed), issue (+ weights), compute per item."""
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
SH = {"blk0": (1, (56, 56), 1, 8), "blk1": (2, (56, 56), 2, 4), "blk2": (2, (28, 28), 1, 4), "blk3": (4, (28, 28), 2, 2),
      "blk4": (4, (14, 14), 1, 2), "blk14": (8, (14, 14), 2, 1), "blk15": (8, (7, 7), 1, 1)}
name = sys.argv[1] if len(sys.argv) > 1 else "blk4"
h, hw, sq, skv = SH[name]
B, n_obj = 128, 4
thw = (1,) + hw
N = 1 + hw[0] * hw[1] + n_obj
g = torch.Generator(device="cpu").manual_seed(1)
qkv = (torch.randn((B, N, 3, h, 96), generator=g) * 0.5).to(DEV, torch.bfloat16)
ws = [(torch.randn((96, 27), generator=g) * 0.2).to(DEV) for _ in range(3)]
gm = [torch.ones(96, device=DEV) for _ in range(3)]
bt = [torch.zeros(96, device=DEV) for _ in range(3)]
J = ops.pooled(hw[0], skv) + ops.pooled(hw[1], skv) + 1
da = 128 if J <= 32 else 160
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(4):
    if i == 3:
        ev0.record()
    ops.pool_ln_fwd_qkv(qkv, ws, gm, bt, B, h, thw, n_obj, (sq, skv, skv), (da, da, 96), (0, 1, 0), save=False)
ev1.record()
torch.cuda.synchronize()
n = 256 * 64
buf = (ctypes.c_ulonglong * n)()
raw = ctypes.CDLL(lib_path)
assert raw.svit_debug_pool_frame_stamps(buf, n) == 0
w = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(256, 64)
t0 = w[:, 0][w[:, 0] > 0].min()
print("%s: launch %.1f us by events" % (name, ev0.elapsed_time(ev1) * 1e3))
row = w[0]
k = 0
while 4 * k + 3 < 64 and row[4 * k + 3] > 0:
    a, b, c, d = row[4 * k:4 * k + 4]
    print("  WG 0 item %2d: top %.1f  wait %.2f  issue %.2f  compute %.2f us" % (k, (a - t0) / 100., (b - a) / 100., (c - b) / 100., (d - c) / 100.))
    k += 1
wait, issue, comp, end = [], [], [], []
for r in w:
    kk = 0
    while 4 * kk + 3 < 64 and r[4 * kk + 3] > 0:
        a, b, c, d = r[4 * kk:4 * kk + 4]
        wait.append((b - a) / 100.); issue.append((c - b) / 100.); comp.append((d - c) / 100.)
        kk += 1
    if kk:
        end.append((r[4 * kk - 1] - t0) / 100.)
print("  all workgroups: items %d | wait med %.2f mean %.2f | issue med %.2f mean %.2f | compute med %.2f mean %.2f | end med %.1f max %.1f us"
      % (len(wait), np.median(wait), np.mean(wait), np.median(issue), np.mean(issue), np.median(comp), np.mean(comp), np.median(end), np.max(end)))
