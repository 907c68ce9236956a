#!/usr/bin/env python3
"""The wide short-K GEMMs of the 14x14 stage (M 13064, K 384) by tile: 128x128 (cfg 4, the round-3 pick), 128x192 (0),
160x256 (9) and 192x192 (10) -- the last two put every workgroup on the chip at once -- and the heuristic (-1);
torch.mm (hipBLASLt) beside them.   python tools/diag/nt_one_round.py   (GPU box)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from svit_amd import hip, ops
from tools.bench_kernels import rnd, timeit, DEV, BF16
lib = hip.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 13064      # 25728 = the frames pass (128 frames x 201 tokens)
for (N, K, epi, tag) in [(1536, 384, hip.EPI_GELU, "fc1+gelu"), (1536, 384, hip.EPI_GELU, "fc1 nosave"), (1536, 384, hip.EPI_DGELU, "fc2-dgrad"),
                         (1536, 384, hip.EPI_BF16, "bf16"), (1152, 384, hip.EPI_BF16, "qkv"),
                         (768, 384, hip.EPI_BF16, "N768"), (1536, 192, hip.EPI_GELU, "K192 gelu")]:
    a, w = rnd(M, K), rnd(N, K)
    bias = torch.zeros(N, device=DEV)
    aux = rnd(M, N) if epi == hip.EPI_DGELU else None
    out = torch.empty(M, N, device=DEV, dtype=BF16)
    out2 = torch.empty(M, N, device=DEV, dtype=BF16) if (epi == hip.EPI_GELU and tag != "fc1 nosave") else None
    res = []
    for cfg in (-1, 4, 0, 9, 10):
        if (cfg == 9 and N % 256) or (cfg in (0, 10) and N % 192) or (cfg == 4 and N % 128):
            res.append("cfg%d: -" % cfg)
            continue
        lib.svit_debug_set(1, cfg)
        us = min(timeit(lambda: ops.gemm_nt(a, w, bias, epi, out=out, out2=out2, aux=aux, save=tag != "fc1 nosave"), iters=40) for _ in range(3))
        res.append("%s:%.1f" % ("auto" if cfg < 0 else "cfg%d" % cfg, us))
    lib.svit_debug_set(1, -1)
    wt = w.t().contiguous()
    us = min(timeit(lambda: torch.mm(a, wt), iters=40) for _ in range(3))
    print("M=%d N=%4d K=%4d %-10s %s  lib(plain):%.1f" % (M, N, K, tag, "  ".join(res), us), flush=True)
