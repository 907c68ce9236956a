"""Object-token share of the pooling-conv weight gradient, isolated: dpre is zero on every patch
row, so dw[c][k] = coef[k] * sum_obj d*x exactly and dw[c][k] / coef[k] must not depend on k.
Quiet, and beside the TN GEMM; library = SVIT_HIP_LIB or the product build."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svit_amd import hip, ops
torch.manual_seed(0)
DEV = "cuda"
B, h, thw, O = 8, 4, (8, 14, 14), 64
L = thw[0] * thw[1] * thw[2]
N = 1 + L + O
qkv = (torch.randn(B, N, 3, h, 96, device=DEV) * 0.5).bfloat16()
a16 = (torch.randn(13064, 384, device=DEV) * 0.5).bfloat16()
tn_out = torch.zeros(384, 384, device=DEV)
ws = torch.empty(8 * 1024 * 1024, device=DEV)
side = torch.cuda.Stream()
print("library:", hip.LIB_PATH)
for s in (2, 1):
    Lo = thw[0] * ops.pooled(thw[1], s) * ops.pooled(thw[2], s)
    Nout = 1 + Lo + O
    dp = torch.zeros(B, h, Nout, 96, device=DEV, dtype=torch.bfloat16)
    dp[:, :, 1 + Lo:] = (torch.randn(B, h, O, 96, device=DEV) * 0.5).bfloat16()
    # expected: G[c] = sum_{b,h,o} d * x ; dw[c][k] = G[c] * nt[kt] nh[ky] nh[kx] / (n_out_t n_out_h^2)
    x = qkv[:, 1 + L:, 1].float().permute(0, 2, 1, 3)           # [B, h, O, 96] (which = 1)
    G = (dp[:, :, 1 + Lo:].float() * x).sum((0, 1, 2)).cpu().double()
    def counts(st):
        n_out = (3 - 1) // st + 1
        n = [0.0, 0.0, 0.0]
        for o in range(n_out):
            for tap in range(3):
                if 0 <= o * st - 1 + tap < 3: n[tap] += 1
        return n, 1.0 / n_out
    nt, ipt = counts(1); nh, iph = counts(s)
    coef = torch.tensor([nt[k // 9] * nh[(k // 3) % 3] * nh[k % 3] for k in range(27)], dtype=torch.float64) * ipt * iph * iph
    want = G[:, None] * coef[None, :]
    for mode in ("quiet", "beside tn"):
        worst = 0.0; bad_taps = set(); nbad = 0
        for it in range(8):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            if mode != "quiet":
                with torch.cuda.stream(side):
                    for _ in range(6): ops.gemm_tn(a16, a16, tn_out)
            dw = torch.zeros(96, 27, device=DEV)
            a = hip.PoolWgradArgs()
            ops._pool_wgrad_args(a, dp, qkv, 1, dw, B, h, thw, O, s, ws)
            hip.call("svit_pool_conv_wgrad", C.byref(a))
            main.wait_stream(side)
            torch.cuda.synchronize()
            err = (dw.cpu().double() - want).abs() / want.abs().max()
            worst = max(worst, float(err.max()))
            idx = (err > 1e-4).nonzero()
            if len(idx):
                nbad += 1
                bad_taps |= set(idx[:, 1].tolist())
        print("stride %d %-10s: worst |dw - expected| / max = %.3g; runs with error > 1e-4: %d/8; taps %s"
              % (s, mode, worst, nbad, sorted(bad_taps)), flush=True)
