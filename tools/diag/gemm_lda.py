"""Diagnostic: does the row stride of A (channel mapping of a tile's rows) matter for svit_gemm_nt?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svit_amd import ops, hip
from gemm_vs_lib import timeit  # noqa

for M, N, K in [(13064, 384, 1536), (13064, 384, 1152), (3656, 768, 3072), (50696, 192, 768),
                (13064, 1536, 384), (13064, 384, 384)]:
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
    b = torch.zeros(N, device="cuda", dtype=torch.float32)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = []
    for pad in (0, 8, 32, 64, 128):
        wide = torch.randn(M, K + pad, device="cuda", dtype=torch.bfloat16)
        a = wide[:, :K]
        res.append("pad%d:%.1f" % (pad, timeit(lambda: ops.gemm_nt(a, w, b, hip.EPI_BF16, out=out))))
    print(f"M {M:6d} N {N:5d} K {K:5d}  " + " ".join(res), flush=True)
