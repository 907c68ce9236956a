#!/usr/bin/env python3
# NOTE (round 5): svit_attn_debug_set(2, m) -- "run only one of the two backward kernels" -- left the product library (it leaves outputs
# unwritten).  Build the two timing variants instead:  python tools/diag/build_variant.py dqonly attn_bwd.hip -DSVIT_DIAG_BWD_ONLY=1  (and =2 for dkv only)
# and run this script with SVIT_HIP_LIB pointing at them; the calls below then return SVIT_ERR_ARG and change nothing.
"""What the rel-pos fold (svit_attn_bwd_args.relD / relR: scatter matrix D, D . R^T on the matrix pipe, added into dq) costs the
dq kernel at the 14x14-stage shape, dq kernel alone (svit_attn_debug_set(2, 2)) and the whole backward.  GPU box."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svit_amd import hip, ops
from svit_amd.engine import rel_sections
from oracle import svit_ref as R
from tools.bench_kernels import rnd, timeit, KSC, BF16, DEV
lib = hip.load()
if lib.svit_attn_debug_set(2, 0) != 0:      # the run-one-backward-kernel switch left every library in round 5
    sys.exit("%s: the loaded library refuses svit_attn_debug_set(2, .): its 'dq only' / 'dkv only' timings would really be both kernels. "
             "Time the -DSVIT_DIAG_BWD_ONLY=1 / =2 variants (tools/diag/build_variant.py) with tools/bench_kernels.py attn instead." % sys.argv[0])
for (B, h, q_thw, k_thw, O) in [(8, 4, (8, 14, 14), (8, 7, 7), 64), (8, 4, (8, 14, 14), (8, 14, 14), 64)]:
    Lq, Lk = q_thw[0] * q_thw[1] * q_thw[2], k_thw[0] * k_thw[1] * k_thw[2]
    Nq, Nk, J = 1 + Lq + O, 1 + Lk + O, sum(k_thw)
    DA = 128 if J <= 32 else 160
    qa, ka, v = rnd(B, h, Nq, DA), (rnd(B, h, Nk, DA).float() * KSC).to(BF16), rnd(B, h, Nk, 96)
    qa[..., 96 + J:] = 0; ka[..., 96 + J:] = 0
    ctx, lse2 = ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=J)
    dctx = rnd(B, Nq, h * 96)
    rows = [2 * max(q_thw[i], k_thw[i]) - 1 for i in (1, 2, 0)]
    offs, lpad = rel_sections(rows)
    idx = [R.rel_index(q_thw[1], k_thw[1]), R.rel_index(q_thw[2], k_thw[2]), R.rel_index(q_thw[0], k_thw[0])]
    kt, kh, kw = k_thw
    body = torch.full((q_thw[0], q_thw[1], q_thw[2], DA - 96), -1, dtype=torch.int32)
    body[..., :kh] = (offs[0] + idx[0].to(torch.int32)).view(1, q_thw[1], 1, kh)
    body[..., kh:kh + kw] = (offs[1] + idx[1].to(torch.int32)).view(1, 1, q_thw[2], kw)
    body[..., kh + kw:J] = (offs[2] + idx[2].to(torch.int32)).view(q_thw[0], 1, 1, kt)
    cmap = torch.full((Nq, DA - 96), -1, dtype=torch.int32)
    cmap[1:1 + Lq] = body.view(Lq, DA - 96)
    cmap = cmap.to(DEV).contiguous()
    rt = (torch.randn(96, lpad, device=DEV) * 0.3).to(BF16)
    out = []
    for only, name in ((2, "dq kernel"), (0, "dq + dkv")):
        lib.svit_attn_debug_set(2, only)
        plain = min(timeit(lambda: ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, 96 ** -0.5, bias_cols=J), iters=30) for _ in range(3))
        dmat = min(timeit(lambda: ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, 96 ** -0.5, bias_cols=J,
                                               reld=(cmap, lpad, 1.4426950408889634, None)), iters=30) for _ in range(3))
        fold = None
        if lpad <= 128:
            fold = min(timeit(lambda: ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, 96 ** -0.5, bias_cols=J,
                                                   reld=(cmap, lpad, 1.4426950408889634, rt, "fold")), iters=30) for _ in range(3))
        out.append("%s: plain %.1f us, + D rows %.1f us, + fold %s us" % (name, plain, dmat, "%.1f" % fold if fold else "-"))
    lib.svit_attn_debug_set(2, 0)
    print("Nq %d Nk %d DA %d lpad %d | " % (Nq, Nk, DA, lpad) + " | ".join(out), flush=True)
