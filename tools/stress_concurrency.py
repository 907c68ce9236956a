import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svit_amd import ops, hip
torch.manual_seed(0)
DEV = "cuda"; dev0 = torch.device("cuda", 0)
B, h, thw, O, sq, skv = 8, 4, (8, 14, 14), 64, 1, 2
N = 1 + thw[0] * thw[1] * thw[2] + O
qkv = (torch.randn(B, N, 3, h, 96, device=DEV) * 0.5).bfloat16()
strides = (sq, skv, skv)
dpres = []
for s in strides:
    Nout = 1 + thw[0] * ops.pooled(thw[1], s) * ops.pooled(thw[2], s) + O
    dpres.append((torch.randn(B, h, Nout, 96, device=DEV) * 0.5).bfloat16())
M = 13064
a16 = (torch.randn(M, 384, device=DEV) * 0.5).bfloat16(); w16 = (torch.randn(1536, 384, device=DEV) * 0.1).bfloat16()
a2 = (torch.randn(4096, 384, device=DEV) * 0.5).bfloat16(); b2 = (torch.randn(4096, 96, device=DEV) * 0.5).bfloat16()
Nq, Nk = 1633, 457
qa = (torch.randn(B, h, Nq, 128, device=DEV) * 0.3).bfloat16(); ka = (torch.randn(B, h, Nk, 128, device=DEV) * 0.3).bfloat16()
vv = (torch.randn(B, h, Nk, 96, device=DEV) * 0.3).bfloat16()
x32 = torch.randn(M, 384, device=DEV); dy32 = torch.randn(M, 384, device=DEV)
g1 = torch.ones(384, device=DEV)
_, _, mean, rstd = ops.layernorm_fwd(x32, g1, torch.zeros(384, device=DEV))
wc = [torch.randn(96, 27, device=DEV) * 0.2 for _ in range(3)]
gs = [torch.ones(96, device=DEV) for _ in range(3)]; bs = [torch.zeros(96, device=DEV) for _ in range(3)]
def partner(kind):
    if kind == "nt": ops.gemm_nt(a16, w16, None, hip.EPI_BF16)
    elif kind == "attn": ops.attn_fwd(qa, ka, vv, 96 ** -0.5)
    elif kind == "tn": ops.gemm_tn(a16[:, :384], a16, torch.zeros(384, 384, device=DEV))
def victim(kind, side_ws):
    if kind == "wgrad3":
        dws = [torch.zeros(96, 27, device=DEV) for _ in range(3)]
        ops.pool_conv_wgrad_qkv(dpres, qkv, dws, B, h, thw, O, strides, ws=side_ws)
        return torch.stack(dws)
    if kind == "wgrad1":
        dw = torch.zeros(96, 27, device=DEV)
        a = hip.PoolWgradArgs(); ops._pool_wgrad_args(a, dpres[0], qkv, 0, dw, B, h, thw, O, 1, side_ws)
        import ctypes as C
        hip.call("svit_pool_conv_wgrad", C.byref(a))
        return dw
    if kind == "wgrad3rows":
        dws = [torch.zeros(96, 27, device=DEV) for _ in range(3)]
        ops.pool_conv_wgrad_qkv(dpres, qkv, dws, B, h, thw, O, strides, ws=side_ws)
        return side_ws[:1024 * 7776].clone()
    if kind == "lnbwd":
        dg, db = torch.zeros(384, device=DEV), torch.zeros(384, device=DEV)
        ops.layernorm_bwd(dy32, x32, g1, mean, rstd, dg, db)
        return torch.cat([dg, db])
    if kind == "tn1":
        dw = torch.zeros(384, 96, device=DEV)
        ops.gemm_tn(a2, b2, dw, splits=1)
        return dw
    if kind == "nt":
        return ops.gemm_nt(a2, w16, None, hip.EPI_BF16)
    if kind == "poolfwd":
        r = ops.pool_ln_fwd_qkv(qkv, wc, gs, bs, B, h, thw, O, strides, (128, 128, 96), (0, 1, 0))
        return torch.cat([x[1].flatten().float() for x in r])
    if kind == "attn":
        return ops.attn_fwd(qa, ka, vv, 96 ** -0.5)[0]
qkv0 = qkv.clone(); dp0 = [d.clone() for d in dpres]
st = torch.cuda.Stream()
side_ws = ops.scratch(dev0, tag="side")
for vk in ("wgrad3rows",):
    for pk in ("nt", "tn"):
        ref = None; bad = 0
        for it in range(30):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                cur = victim(vk, side_ws)
            for _ in range(6): partner(pk)
            torch.cuda.current_stream().wait_stream(st)
            torch.cuda.synchronize()
            if ref is None: ref = cur.clone()
            elif not torch.equal(cur, ref):
                bad += 1
                if bad <= 3 and cur.numel() == 1024 * 7776:
                    d = (cur - ref).view(1024, 3, 2592).abs()
                    rows = (d.amax(dim=2) > 0).nonzero()
                    print("   differing (block, which):", rows[:8].tolist(), "n=", len(rows), "max diff %.3g" % float(d.max()),
                          "ref scale %.3g" % float(ref.abs().max()))
        print("victim %-8s partner %-5s: %d/30 mismatching" % (vk, pk, bad),
              "inputs intact:", torch.equal(qkv, qkv0), [torch.equal(a, b) for a, b in zip(dpres, dp0)])
