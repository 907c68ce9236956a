"""one q/k/v pooling forward shape in a loop (for rocprofv3 --pmc): pool_one.py h T H W sq skv [sel]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svit_amd import ops, hip
h, T, H, W, sq, skv = (int(a) for a in sys.argv[1:7])
use_sel = len(sys.argv) > 7 and sys.argv[7] == "sel"
B, n_obj, DEV = 8, 64, "cuda"
thw = (T, H, W)
N = 1 + T * H * W + n_obj
qkv = torch.randn(B, N, 3, h, 96, device=DEV).bfloat16()
ws = [torch.randn(96, 27, device=DEV) * 0.2 for _ in range(3)]
g = [torch.ones(96, device=DEV) for _ in range(3)]
b = [torch.zeros(96, device=DEV) for _ in range(3)]
wflat = torch.cat([w.flatten() for w in ws]).contiguous()
offs = torch.tensor([0, 2592, 5184], dtype=torch.int64, device=DEV)
sel = ops.pool_weight_sel(wflat, offs, torch.zeros((3, 2592), dtype=torch.int32, device=DEV))
sels = [sel[i] for i in range(3)] if use_sel else None
J = 2 * ops.pooled(H, skv) + T
da = 128 if J <= 32 else 160
for _ in range(6):
    ops.pool_ln_fwd_qkv(qkv, ws, g, b, B, h, thw, n_obj, (sq, skv, skv), (da, da, 96), (0, 1, 0), sels=sels)
torch.cuda.synchronize()
