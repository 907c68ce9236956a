"""Second-level diagnostic (see tools/stress_concurrency.py): which victim kernel variants and
which partner kernels reproduce the differing partial rows, and what the differing elements are."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from svit_amd import hip, ops

torch.manual_seed(0)

def _streaming():
    """the streaming conv-backward wrappers (diagnostic build only since round 6: tools/diag/pool_streaming.py)"""
    from tools.diag import pool_streaming
    return pool_streaming


DEV = "cuda"
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B, h, thw, O = 8, 4, (8, 14, 14), 64
N = 1 + thw[0] * thw[1] * thw[2] + O
qkv = (torch.randn(B, N, 3, h, 96, device=DEV) * 0.5).bfloat16()


def mk_dpre(s):
    Nout = 1 + thw[0] * ops.pooled(thw[1], s) * ops.pooled(thw[2], s) + O
    return (torch.randn(B, h, Nout, 96, device=DEV) * 0.5).bfloat16()


dp = {1: mk_dpre(1), 2: mk_dpre(2)}
M = 13064
a16 = (torch.randn(M, 384, device=DEV) * 0.5).bfloat16()
tn_out = torch.zeros(384, 384, device=DEV)
tn_out2 = torch.zeros(384, 96, device=DEV)
colsum_out = torch.zeros(384, device=DEV)
idx = torch.randint(0, 384 * 384, (4 * 1024 * 1024,), device=DEV)
src = torch.randn(4 * 1024 * 1024, device=DEV)
acc_t = torch.zeros(384 * 384, device=DEV)
x32 = torch.randn(M, 384, device=DEV)
g1 = torch.ones(384, device=DEV)
_, _, mean, rstd = ops.layernorm_fwd(x32, g1, torch.zeros(384, device=DEV))
dg, db = torch.zeros(384, device=DEV), torch.zeros(384, device=DEV)
lnws = torch.empty(8 * 1024 * 1024, device=DEV)


def partner(kind):
    if kind == "tn":
        ops.gemm_tn(a16, a16, tn_out)
    elif kind == "tn_split1":
        ops.gemm_tn(a16, a16, tn_out, splits=1)
    elif kind == "tn_grouped":
        ops.gemm_tn_grouped([(a16, a16, tn_out, None), (a16, a16[:, :96], tn_out2, None)])
    elif kind == "colsum":
        ops.colsum(a16, colsum_out)
    elif kind == "scatter_add":          # global fp32 atomics only (no LDS, no MFMA)
        acc_t.index_add_(0, idx, src)
    elif kind == "lnbwd":
        ops.layernorm_bwd(x32, x32, g1, mean, rstd, dg, db, ws=lnws)


def victim(kind, ws):
    ws.fill_(-7.0)
    if kind == "qkv122":       # fused launch, strides (1, 2, 2)
        dws = [torch.zeros(96, 27, device=DEV) for _ in range(3)]
        _streaming().pool_conv_wgrad_qkv([dp[1], dp[2], dp[2]], qkv, dws, B, h, thw, O, (1, 2, 2), ws=ws)
        return ws[:1024 * 7776].clone(), 7776
    if kind == "qkv111":
        dws = [torch.zeros(96, 27, device=DEV) for _ in range(3)]
        _streaming().pool_conv_wgrad_qkv([dp[1], dp[1], dp[1]], qkv, dws, B, h, thw, O, (1, 1, 1), ws=ws)
        return ws[:1024 * 7776].clone(), 7776
    s = 2 if kind == "single_s2" else 1      # the single-tensor kernel
    dw = torch.zeros(96, 27, device=DEV)
    a = hip.PoolWgradArgs()
    ops._pool_wgrad_args(a, dp[s], qkv, 1, dw, B, h, thw, O, s, ws)
    hip.call("svit_pool_conv_wgrad", C.byref(a))
    return ws[:1024 * 2592].clone(), 2592


side = torch.cuda.Stream()
ws = torch.empty(8 * 1024 * 1024, device=DEV)
torch.cuda.synchronize()
for vk in ("qkv122", "qkv111", "single_s2", "single_s1"):
    for pk in ("tn", "tn_split1", "tn_grouped", "colsum", "scatter_add", "lnbwd"):
        ref = None
        bad = 0
        notes = []
        for it in range(RUNS):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(6):
                    partner(pk)
            rows, width = victim(vk, ws)
            main.wait_stream(side)
            torch.cuda.synchronize()
            if ref is None:
                ref = rows.clone()
                continue
            if not torch.equal(rows, ref):
                bad += 1
                if len(notes) < 2:
                    d = (rows - ref).view(-1, width)
                    r = d.abs().amax(dim=1).nonzero().flatten()
                    r0 = int(r[0])
                    e = d[r0].nonzero().flatten()
                    sec = e // 2592
                    cc = (e % 2592) // 27
                    tap = e % 27
                    notes.append("run %d: %d rows differ; row %d: %d elements differ, sections %s, channels %d..%d "
                                 "(%d distinct), taps %s; got/ref of first: %.5g / %.5g; diffs %s"
                                 % (it, len(r), r0, len(e), sorted(set(sec.tolist())), int(cc.min()), int(cc.max()),
                                    len(set(cc.tolist())), sorted(set(tap.tolist()))[:27],
                                    float(rows.view(-1, width)[r0, e[0]]), float(ref.view(-1, width)[r0, e[0]]),
                                    [round(float(v), 4) for v in d[r0, e[:6]]]))
        print("victim %-10s partner %-11s: rows differ in %d/%d runs" % (vk, pk, bad, RUNS - 1), flush=True)
        for n in notes:
            print("      ", n)
