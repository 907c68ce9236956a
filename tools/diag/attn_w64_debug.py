#!/usr/bin/env python3
"""svit_attn_fwd through the 64-rows-per-wave kernel (SVIT_ATTN_FWD_W64=1) against the fp32 formula, error by query row / column."""
import math, os, sys
os.environ["SVIT_ATTN_FWD_W64"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svit_amd import ops
torch.manual_seed(0)
for (Nq, Nk, DA, h, J) in [(200, 70, 128, 2, 0), (130, 300, 160, 2, 0), (130, 300, 160, 2, 40), (130, 256, 160, 2, 0), (130, 64, 160, 1, 0), (700, 129, 160, 1, 0), (1633, 1633, 160, 4, 36)]:
    B = 2
    qa = (torch.randn(B, h, Nq, DA, device="cuda") * 0.5).bfloat16()
    ka = (torch.randn(B, h, Nk, DA, device="cuda") * 0.15).bfloat16()
    v = (torch.randn(B, h, Nk, 96, device="cuda") * 0.5).bfloat16()
    if J:
        qa[..., 96 + J:] = 0; ka[..., 96 + J:] = 0
    ctx, lse2 = ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=J)
    s = (qa.float() @ ka.float().transpose(-1, -2)) * math.log(2.0)
    o = s.softmax(-1) @ v.float()
    o = torch.cat([o[:, :, :1], o[:, :, 1:] + qa[:, :, 1:, :96].float()], dim=2)
    ref = o.transpose(1, 2).reshape(B, Nq, h * 96)
    err = (ctx.float() - ref).abs()
    lref = torch.logsumexp(s, -1) / math.log(2.0)
    lerr = (lse2 - lref).abs()
    bad_rows = (err.amax(dim=(0, 2)) > 0.05).nonzero().flatten().tolist()
    print("Nq %d Nk %d DA %d h %d J %d: max err %.4f (ref max %.2f)  lse err %.4f  bad rows %d %s  nan %d"
          % (Nq, Nk, DA, h, J, float(err.max()), float(ref.abs().max()), float(lerr.max()), len(bad_rows), bad_rows[:12], int(torch.isnan(ctx.float()).sum())), flush=True)
    if bad_rows:
        r = bad_rows[0]
        e = err[0, r].view(h, 96)
        print("   row %d batch 0: per-head max err %s; first cols of head 0: got %s ref %s" % (r, e.amax(1).tolist(), ctx[0, r, :6].float().tolist(), ref[0, r, :6].tolist()))
