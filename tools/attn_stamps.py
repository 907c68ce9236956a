"""Cycle anatomy of the attention forward tile loop (workgroup 0, wave 0): builds attn_fwd.hip with
-DSVIT_ATTN_STAMPS into gpurun_out/, runs one shape and prints the per-phase cycle counts.

    python tools/attn_stamps.py [Nq Nk DA heads]
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
    Nq, Nk, DA, h = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (6337, 1633, 160, 2)
    B = 8
    out = os.path.join(ROOT, "gpurun_out", "libattn_stamps.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17",
                           "-ffast-math", "-fno-finite-math-only", "-fno-slp-vectorize", "-DSVIT_ATTN_STAMPS", "-shared",
                           os.path.join(ROOT, "svit_amd", "csrc", "attn_fwd.hip"), "-o", out])
    lib = ctypes.CDLL(out)

    class Args(ctypes.Structure):
        _fields_ = [(n, ctypes.c_void_p) for n in ("qa", "ka", "v", "ctx", "lse2")] + \
                   [(n, ctypes.c_int32) for n in ("B", "heads", "Nq", "Nk", "DA")] + [("scale", ctypes.c_float),
                                                                            ("bias_cols", ctypes.c_int32)]
    dev = "cuda"
    qa = (torch.randn(B, h, Nq, DA, device=dev) * 0.5).bfloat16()
    ka = (torch.randn(B, h, Nk, DA, device=dev) * 0.5).bfloat16()
    v = torch.randn(B, h, Nk, 96, device=dev).bfloat16()
    ctx = torch.empty(B, Nq, h * 96, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, h, Nq, device=dev)
    a = Args(qa.data_ptr(), ka.data_ptr(), v.data_ptr(), ctx.data_ptr(), lse.data_ptr(), B, h, Nq, Nk, DA, 96 ** -0.5,
             22 if DA == 128 else 36)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        rc = lib.svit_attn_fwd(ctypes.byref(a), st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    n = 8192
    buf = (ctypes.c_ulonglong * n)()
    assert lib.svit_debug_attn_stamps(buf, n) == 0
    s = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    nt = (Nk + 63) // 64
    cyc, wall = s[2] - s[0], (s[3] - s[1]) / 100e6
    print("shape Nq=%d Nk=%d DA=%d h=%d: loop %d cycles in %.2f us -> %.2f GHz, %d tiles, %.0f cycles/tile"
          % (Nq, Nk, DA, h, cyc, wall * 1e6, cyc / wall / 1e9, nt, cyc / nt))
    print("prologue (entry -> loop) %d cycles, epilogue (loop end -> stores retired) %d cycles"
          % (s[0] - s[4], s[5] - s[2]))
    names = ["vmcnt wait", "barrier", "dma issue", "QK^T", "softmax", "PV"]
    rows = []
    for t in range(nt):
        b = 8 + t * 8
        prev_end = s[8 + (t - 1) * 8 + 4] if t else s[0]
        rows.append([s[b + 5] - prev_end, s[b + 0] - s[b + 5], s[b + 1] - s[b + 0], s[b + 2] - s[b + 1],
                     s[b + 3] - s[b + 2], s[b + 4] - s[b + 3]])
    rows = np.array(rows)
    print("per-tile cycles (median over tiles 1..%d | tile 0 | last):" % (nt - 2))
    for i, nme in enumerate(names):
        print("  %-11s %7.0f | %7d | %7d" % (nme, np.median(rows[1:-1, i]) if nt > 2 else rows[0, i], rows[0, i], rows[-1, i]))
    print("  %-11s %7.0f" % ("sum", np.median(rows[1:-1].sum(1)) if nt > 2 else rows.sum()))


if __name__ == "__main__":
    main()
