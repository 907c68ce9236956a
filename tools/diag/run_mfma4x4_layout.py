#!/usr/bin/env python3
"""Checks the operand layout of v_mfma_f32_4x4x4_16b_bf16 assumed by the MFMA pooling stencil (csrc/pool.hip,
pool_mfma_fwd_kernel) with exact integer data.  GPU box: python tools/diag/run_mfma4x4_layout.py"""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib_path = os.path.join(here, "libmfma4x4_layout.so")
if not os.path.exists(lib_path):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(here, "mfma4x4_layout.hip"), "-o", lib_path])
lib = ctypes.CDLL(lib_path)
g = torch.Generator().manual_seed(1)
A = torch.randint(-4, 5, (16, 4, 4), generator=g).float()      # [block][i][k]
Bm = torch.randint(-4, 5, (16, 4, 4), generator=g).float()     # [block][k][j]
D = torch.einsum("bik,bkj->bij", A, Bm)
a_l = torch.zeros(64, 4); b_l = torch.zeros(64, 4)
for blk in range(16):
    for x in range(4):
        a_l[4 * blk + x] = A[blk, x, :]          # lane 4b+i holds A[i][0..3]
        b_l[4 * blk + x] = Bm[blk, :, x]         # lane 4b+j holds B[0..3][j]
a_d = a_l.to(torch.bfloat16).cuda().view(torch.int16)
b_d = b_l.to(torch.bfloat16).cuda().view(torch.int16)
d_d = torch.zeros(64, 4, device="cuda")
rc = lib.mfma4x4_probe(ctypes.c_void_p(a_d.data_ptr()), ctypes.c_void_p(b_d.data_ptr()), ctypes.c_void_p(d_d.data_ptr()), None)
torch.cuda.synchronize()
got = d_d.cpu()
cand = {"lane=4b+j, reg=i": torch.stack([torch.stack([D[b, :, j] for j in range(4)]) for b in range(16)]).reshape(64, 4),
        "lane=4b+i, reg=j": torch.stack([torch.stack([D[b, i, :] for i in range(4)]) for b in range(16)]).reshape(64, 4)}
ok = False
for name, want in cand.items():
    m = bool(torch.equal(got, want))
    ok |= m
    print("D layout '%s': %s" % (name, "MATCH" if m else "no"))
print("rc", rc, "layout confirmed" if ok else "NO candidate matched: lane 0..7 got %s" % got[:8].tolist())
