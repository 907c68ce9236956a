"""Cycle anatomy of the anti-phase attention forward (csrc/attn_fwd.hip::attn_fwd_ap_kernel; workgroup 0, waves 0 and 4):
builds attn_fwd.hip with -DSVIT_ATTN_STAMPS (tools/diag/build_variant.py), runs one shape with the kernel forced on and
prints the per-segment cycle counts of both halves.     python tools/attn_ap_stamps.py [Nq Nk DA heads]   (GPU box)"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    Nq, Nk, DA, h = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (1633, 1633, 160, 4)
    mask = int(sys.argv[5], 0) if len(sys.argv) > 5 else 255      # which of the 8 per-tile stamp points exist (0 = loop totals only)
    extra = sys.argv[6:]
    tag = "apstamps%d" % mask + "".join(e.replace("-D", "_").replace("=", "") for e in extra)
    lib_path = os.path.join(ROOT, "tools", "diag", "libsvit_diag_%s.so" % tag)
    if not os.path.exists(lib_path):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "diag", "build_variant.py"), tag, "attn_fwd.hip",
                               "-DSVIT_ATTN_STAMPS=%d" % mask] + extra)
    os.environ["SVIT_HIP_LIB"] = lib_path
    import torch
    from svit_amd import hip, ops
    lib = hip.load()
    B = 8
    J = 22 if DA == 128 else 36
    qa = (torch.randn(B, h, Nq, DA, device="cuda") * 0.5).bfloat16()
    ka = (torch.randn(B, h, Nk, DA, device="cuda") * 0.5 * 0.1472).bfloat16()
    v = torch.randn(B, h, Nk, 96, device="cuda").bfloat16()
    qa[..., 96 + J:] = 0
    ka[..., 96 + J:] = 0
    assert lib.svit_attn_debug_set(4, 1) == 0 and lib.svit_attn_debug_set(5, 0) == 0
    for _ in range(3):
        ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=J)
    torch.cuda.synchronize()
    raw = ctypes.CDLL(lib_path)
    n = 2 * 64 * 8 + 8
    buf = (ctypes.c_ulonglong * n)()
    assert raw.svit_debug_attn_ap_stamps(buf, n) == 0
    s = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    nt = (Nk + 63) // 64
    cyc, wall = s[1026] - s[1024], (s[1027] - s[1025]) / 100e6
    print("shape Nq=%d Nk=%d DA=%d h=%d B=%d: entry -> loop end %d cycles in %.2f us -> %.2f GHz, %d tiles; stores retired %d cycles later"
          % (Nq, Nk, DA, h, B, cyc, wall * 1e6, cyc / wall / 1e9, nt, s[1028] - s[1026]))
    print("   per tile: %.0f cycles" % ((s[1026] - s[1024]) / nt))
    if mask == 0:
        return
    st = s[:1024].reshape(2, 64, 8)
    if mask != 255:
        pts = [i for i in range(8) if (mask >> i) & 1]
        for half in (0, 1):
            sub = st[half, 2:min(nt, 64) - 2][:, pts]
            d = np.diff(np.concatenate([sub, np.roll(sub[:, :1], -1, axis=0)], axis=1), axis=1)[:-1]
            print("half %d: stamp points %s -> median deltas %s (sum %.0f)" % (half, pts, [int(x) for x in np.median(d, axis=0)],
                                                                             np.median(d.sum(1))))
        print("half 1's point %d comes %.0f cycles after half 0's (median)" % (pts[0], np.median(st[1, 2:nt - 2, pts[0]] - st[0, 2:nt - 2, pts[0]])))
        return
    names = ["P.V(t-1)", "QK^T(t)", "vmcnt wait", "barrier", "softmax", "DMA issue", "barrier"]
    lo, hi = 2, min(nt, 64) - 2
    for half in (0, 1):
        d = np.diff(st[half, lo:hi], axis=1)
        per = st[half, lo + 1:hi, 0] - st[half, lo:hi - 1, 0]
        print("half %d (wave %d): median cycles over tiles %d..%d; tile period %.0f" % (half, 4 * half, lo, hi - 1, np.median(per)))
        for i, nme in enumerate(names):
            print("   %-11s %6.0f   (min %5d max %5d)" % (nme, np.median(d[:, i]), d[:, i].min(), d[:, i].max()))
        print("   matrix segment %.0f (+ wait/barrier %.0f), vector segment %.0f (+ barrier %.0f)" % (
            np.median(d[:, 0] + d[:, 1]), np.median(d[:, 2] + d[:, 3]), np.median(d[:, 4] + d[:, 5]), np.median(d[:, 6])))
    # phase relation: half 1's matrix segment start minus half 0's
    print("half 1 starts its matrix segment %.0f cycles after half 0 (median)" % np.median(st[1, lo:hi, 0] - st[0, lo:hi, 0]))


if __name__ == "__main__":
    main()
