"""single-tensor pooling wgrad (stride 2 and 1) beside the TN GEMM, with the library named by
SVIT_HIP_LIB (normal build vs a build whose two row halves met in global memory instead of LDS --
the diagnostic macro that produced it was removed from pool.hip after the root cause was found)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svit_amd import hip, ops
torch.manual_seed(0)
DEV = "cuda"
B, h, thw, O = 8, 4, (8, 14, 14), 64
N = 1 + thw[0] * thw[1] * thw[2] + O
qkv = (torch.randn(B, N, 3, h, 96, device=DEV) * 0.5).bfloat16()
def mk(s):
    Nout = 1 + thw[0] * ops.pooled(thw[1], s) * ops.pooled(thw[2], s) + O
    return (torch.randn(B, h, Nout, 96, device=DEV) * 0.5).bfloat16()
dp = {1: mk(1), 2: mk(2)}
a16 = (torch.randn(13064, 384, device=DEV) * 0.5).bfloat16()
tn_out = torch.zeros(384, 384, device=DEV)
ws = torch.empty(8 * 1024 * 1024, device=DEV)
side = torch.cuda.Stream()
print("library:", hip.LIB_PATH)
for s in (2, 1):
    for n_obj in (O, 0):
        ref = None; bad = 0
        for it in range(12):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(6): ops.gemm_tn(a16, a16, tn_out)
            ws.fill_(-7.0)
            dw = torch.zeros(96, 27, device=DEV)
            a = hip.PoolWgradArgs()
            ops._pool_wgrad_args(a, dp[s], qkv, 1, dw, B, h, thw, n_obj, s, ws)
            if n_obj == 0:     # same tensors, object rows simply not visited
                pass
            hip.call("svit_pool_conv_wgrad", C.byref(a))
            rows = ws[:1024 * 2592].clone()
            main.wait_stream(side)
            torch.cuda.synchronize()
            if ref is None: ref = rows.clone()
            elif not torch.equal(rows, ref): bad += 1
        print("stride %d n_obj %2d: rows differ in %d/11 runs" % (s, n_obj, bad), flush=True)
