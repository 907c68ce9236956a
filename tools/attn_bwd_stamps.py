"""Wall-clock anatomy (100 MHz timer) of workgroup 0 / wave 0 of the attention backward kernels:
builds attn_bwd.hip with -DSVIT_ATTN_STAMPS into gpurun_out/, runs one shape, prints the timeline.

    python tools/attn_bwd_stamps.py [Nq Nk DA heads halves splits [fold]]
(7th argument 1: with the rel-pos backward folded into the dq epilogue -- relD / relR, synthetic map -- so that
"epilogue" shows what the scatter + D.R^T + fold cost)
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    Nq, Nk, DA, h, halves, splits = [int(v) for v in sys.argv[1:7]] if len(sys.argv) >= 7 else (1633, 457, 128, 4, 0, 0)
    B = 8
    out = os.path.join(ROOT, "gpurun_out", "libattn_bwd_stamps.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17",
                           "-ffast-math", "-fno-finite-math-only", "-fno-slp-vectorize", "-DSVIT_ATTN_STAMPS", "-shared",
                           os.path.join(ROOT, "svit_amd", "csrc", "attn_bwd.hip"), "-o", out])
    lib = ctypes.CDLL(out)
    from svit_amd import ops, hip

    dev = "cuda"
    J = 22 if DA == 128 else 36
    qa = (torch.randn(B, h, Nq, DA, device=dev) * 0.5).bfloat16()
    ka = (torch.randn(B, h, Nk, DA, device=dev) * 0.5 * 0.1472).bfloat16()
    v = torch.randn(B, h, Nk, 96, device=dev).bfloat16()
    ctx, lse2 = ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=J)
    dctx = torch.randn(B, Nq, h * 96, device=dev).bfloat16()
    dqa = torch.empty_like(qa)
    delta = torch.empty(B, h, Nq, 2, device=dev)
    a = hip.AttnBwdArgs()
    a.qa, a.ka, a.v, a.ctx, a.dctx, a.lse2 = (t.data_ptr() for t in (qa, ka, v, ctx, dctx, lse2))
    a.delta, a.dqa = delta.data_ptr(), dqa.data_ptr()
    a.B, a.heads, a.Nq, a.Nk, a.DA, a.q_splits, a.scale, a.bias_cols = B, h, Nq, Nk, DA, splits, 96 ** -0.5, J
    if len(sys.argv) > 7 and sys.argv[7] == "1":
        ldd = 96
        jj = torch.arange(DA - 96)
        cmap = ((jj[None, :] * 3 + torch.arange(Nq)[:, None]) % 69).to(torch.int32)
        cmap[:, J:] = -1
        cmap[0] = -1
        cmap = cmap.contiguous().to(dev)
        D = torch.empty(B * h * Nq, ldd, device=dev, dtype=torch.bfloat16)
        rt = (torch.randn(96, ldd, device=dev) * 0.1).bfloat16()
        a.relD, a.relD_ld, a.relD_map, a.relD_scale, a.relR = D.data_ptr(), ldd, cmap.data_ptr(), 1.4426950408889634, rt.data_ptr()
        print("(rel-pos backward folded into the dq epilogue: relD / relR, ldd 96)")
    lib.svit_attn_debug_set(0, halves)
    parts = lib.svit_attn_bwd_parts(ctypes.byref(a))
    assert parts >= 1, parts
    dkv = torch.empty(2, parts, B, h, Nk, 96, device=dev)
    a.dk, a.dv, a.q_splits = dkv[0].data_ptr(), dkv[1].data_ptr(), parts
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        rc = lib.svit_attn_bwd(ctypes.byref(a), st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    assert lib.svit_debug_attn_bwd_stamps(buf, 64) == 0
    s = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    us = lambda i, j: (s[j] - s[i]) / 100.0
    print("shape Nq=%d Nk=%d DA=%d h=%d halves=%d splits=%d (us, workgroup 0 / wave 0)" % (Nq, Nk, DA, h, halves, splits))
    print("dq : entry->issue0 %.2f | issue0 %.2f | operands+delta+pin %.2f | loop %.2f | epilogue %.2f | total %.2f"
          % (us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(0, 5)))
    for t in range(min(4, (Nk + 63) // 64)):
        b = 8 + 4 * t
        print("   tile %d: vmcnt wait %.2f  barrier %.2f  dma issue %.2f  compute(to next tile) %.2f"
              % (t, us(b, b + 1), us(b + 1, b + 2), us(b + 2, b + 3), us(b + 3, b + 4) if t < 3 and s[b + 4] else float("nan")))
    print("dkv: entry->issue %.2f | issue %.2f | operands+pin %.2f | loop %.2f | halves merge %.2f | stores %.2f | total %.2f (parts %d)"
          % (us(32, 33), us(33, 34), us(34, 35), us(35, 36), us(36, 37), us(37, 38), us(32, 38), parts))
    for t in range(4):
        b = 44 + 4 * t
        if s[b] == 0:
            break
        print("   stage %d: vmcnt wait %.2f  barrier %.2f  dma issue %.2f  compute %.2f"
              % (t, us(b, b + 1), us(b + 1, b + 2), us(b + 2, b + 3), us(b + 3, b + 4) if t < 3 and s[b + 4] else float("nan")))


if __name__ == "__main__":
    main()
