"""Diagnostic: fixed cost (launch + prologue + epilogue) of svit_gemm_nt: time against K at fixed M, N and epilogue."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svit_amd import ops, hip
from gemm_vs_lib import timeit  # noqa
import ctypes as C
lib = hip.load()
lib.svit_debug_set.restype, lib.svit_debug_set.argtypes = C.c_int32, [C.c_int32, C.c_int32]
M = int(os.environ.get("NT_M", "13064"))
for N in (384, 1152, 1536) if M < 100000 else (96, 288, 384, 576):
    for epi, name in ((hip.EPI_BF16, "bf16"), (hip.EPI_GELU, "gelu"), (hip.EPI_RESID, "resid"), (hip.EPI_F32, "f32")):
        res = []
        for K in ((64, 128, 384, 768, 1536) if M < 100000 else (96, 192, 384)):
            a = torch.randn(M, K, device="cuda").bfloat16()
            w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
            b = torch.zeros(N, device="cuda")
            aux = torch.randn(M, N, device="cuda") if epi == hip.EPI_RESID else None
            out = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi in (hip.EPI_RESID, hip.EPI_F32) else torch.bfloat16)
            out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == hip.EPI_GELU else None
            t = timeit(lambda: ops.gemm_nt(a, w, b, epi, out=out, out2=out2, aux=aux))
            res.append("K%d:%.1f" % (K, t))
        print("M %d N %4d %-5s  %s" % (M, N, name, "  ".join(res)), flush=True)
