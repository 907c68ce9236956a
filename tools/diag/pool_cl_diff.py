"""Diagnostic: where does the channel-lane forward differ from the streaming kernel?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svit_amd import ops, hip
hip.load()
DEV = "cuda"
torch.manual_seed(1)
sq, skv, thw = int(sys.argv[1]), int(sys.argv[2]), tuple(int(a) for a in sys.argv[3:6])
B, h, O = 2, 2, 3
N = 1 + thw[0] * thw[1] * thw[2] + O
if len(sys.argv) > 6:      # the unit test's own inputs
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
    import test_kernels_gpu as TK
    qkv = TK._qkv(B, h, thw, O, "t%d%d%d" % (sq, skv, thw[1]))
    ws = [TK.rnd("tw%d" % i, (96, 27), 0.3) for i in range(3)]
    gs = [TK.rnd("tg%d" % i, (96,), 0.2) + 1.0 for i in range(3)]
    bs = [TK.rnd("tb%d" % i, (96,), 0.1) for i in range(3)]
else:
    qkv = (torch.randn(B, N, 3, h, 96, device=DEV)).bfloat16()
    ws = [torch.randn(96, 27, device=DEV) * 0.3 for _ in range(3)]
    gs = [torch.ones(96, device=DEV) for _ in range(3)]
    bs = [torch.zeros(96, device=DEV) for _ in range(3)]
strides, lds, modes = (sq, skv, skv), (160, 160, 96), (0, 1, 0)
wflat = torch.cat([w.flatten() for w in ws]).contiguous()
offs = torch.tensor([0, 2592, 5184], dtype=torch.int64, device=DEV)
sel = ops.pool_weight_sel(wflat, offs, torch.zeros((3, 2592), dtype=torch.int32, device=DEV))
sels = [sel[i] for i in range(3)]
ref = ops.pool_ln_fwd_qkv(qkv, ws, gs, bs, B, h, thw, O, strides, lds, modes)
got = ops.pool_ln_fwd_qkv(qkv, ws, gs, bs, B, h, thw, O, strides, lds, modes, sels=sels)
torch.cuda.synchronize()
for i in range(3):
    s = strides[i]
    Ho, Wo = ops.pooled(thw[1], s), ops.pooled(thw[2], s)
    a, b = got[i][1].float(), ref[i][1].float()     # pre [B, h, Nout, 96]
    d = (a - b).abs()
    bad = (d > 1e-6).nonzero()
    print("tensor %d stride %d: %d of %d elements differ, max diff %.4f, max ref %.3f" %
          (i, s, bad.shape[0], d.numel(), float(d.max()), float(b.abs().max())))
    if bad.shape[0]:
        toks = bad[:, 2].unique()
        rows = []
        for t in toks[:40].tolist():
            if t == 0 or t > thw[0] * Ho * Wo:
                rows.append("special%d" % t)
            else:
                p = t - 1
                rows.append("(t%d y%d x%d)" % (p // (Ho * Wo), (p // Wo) % Ho, p % Wo))
        print("   tokens:", len(toks), rows)
        chs = bad[:, 3].unique().tolist()
        print("   channels:", chs[:40], "bh:", bad[:, 0].unique().tolist(), bad[:, 1].unique().tolist())
        k = bad[0].tolist()
        print("   first:", k, float(a[tuple(k)]), float(b[tuple(k)]))
