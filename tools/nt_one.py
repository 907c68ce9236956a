"""one NT GEMM shape in a loop (for rocprofv3 --pmc): python tools/nt_one.py M N K [epi] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svit_amd import ops, hip
M, N, K = (int(a) for a in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.1).bfloat16()
aux = torch.randn(M, N, device="cuda") if epi == 2 else None
for _ in range(iters):
    ops.gemm_nt(a, w, None, epi, aux=aux)
torch.cuda.synchronize()
