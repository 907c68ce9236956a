"""Diagnostic: which library kernels torch.mm picks for the step's dominant GEMM shapes (run under
rocprofv3 --kernel-trace --stats; the kernel names carry the macro-tile / split configuration)."""
import torch
for M, N, K in [(13064, 384, 1536), (13064, 384, 1152), (3656, 768, 3072), (50696, 192, 768), (13064, 1536, 384)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(10):
        torch.mm(a, w.t(), out=out)
    torch.cuda.synchronize()
