import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svit_amd import hip, ops
from tools.bench_kernels import timeit, rnd, B
DEV = "cuda"
print("lib:", os.path.basename(hip.LIB_PATH))
for blk, h, thw, sq, skv in [(0, 1, (8, 56, 56), 1, 8), (4, 4, (8, 14, 14), 1, 2)]:
    n_obj = 64
    N = 1 + thw[0] * thw[1] * thw[2] + n_obj
    qkv = rnd(B, N, 3, h, 96)
    ws = [torch.randn(96, 27, device=DEV) * 0.2 for _ in range(3)]
    g = [torch.ones(96, device=DEV) for _ in range(3)]
    b = [torch.zeros(96, device=DEV) for _ in range(3)]
    sel = torch.zeros((3, 2592), dtype=torch.int32, device=DEV)
    sels = [sel[i] for i in range(3)]
    J = 2 * ops.pooled(thw[1], skv) + thw[0]
    da = 128 if J <= 32 else 160
    strides, lds, modes = (sq, skv, skv), (da, da, 96), (0, 1, 0)
    f0 = timeit(lambda: ops.pool_ln_fwd_qkv(qkv, ws, g, b, B, h, thw, n_obj, strides, lds, modes))
    f1 = timeit(lambda: ops.pool_ln_fwd_qkv(qkv, ws, g, b, B, h, thw, n_obj, strides, lds, modes, sels=sels))
    print("blk%d fwd streaming %.1f us | tiled %.1f us" % (blk, f0, f1), flush=True)
