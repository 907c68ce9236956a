#!/usr/bin/env python3
"""Per-shape micro-benchmarks of the hot kernels at the BASELINE C2 shapes (B=8, 16x224^2).
Usage (GPU box): python tools/bench_kernels.py [gemm|tn|attn|pool|all]"""
import json
import os
import sys


def _streaming():
    """the streaming conv-backward wrappers (diagnostic build only since round 6: tools/diag/pool_streaming.py)"""
    from tools.diag import pool_streaming
    return pool_streaming


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from svit_amd import hip, ops

DEV = "cuda"
BF16 = torch.bfloat16
B = 8
# (blk, N_in, Nq, Nk, C_in, C_out, heads, DA)
BLOCKS = [(0, 25153, 25153, 457, 96, 96, 1, 128), (1, 25153, 6337, 1633, 96, 192, 2, 160),
          (2, 6337, 6337, 457, 192, 192, 2, 128), (3, 6337, 1633, 1633, 192, 384, 4, 160),
          (4, 1633, 1633, 457, 384, 384, 4, 128), (14, 1633, 457, 1633, 384, 768, 8, 160),
          (15, 457, 457, 457, 768, 768, 8, 128)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


KSC = (96 ** -0.5) * 1.4426950408889634   # the pooled keys carry scale * log2(e) (engine.K_SCALE)


def rnd(*shape, dtype=BF16):
    return (torch.randn(*shape, device=DEV) * 0.5).to(dtype)


def bench_gemm():
    print("== gemm_nt (M, N, K, epilogue) ==")
    seen = set()
    for blk, Nin, Nq, Nk, Ci, Co, h, DA in BLOCKS:
        for (M, N, K, epi, tag) in [(B * Nin, 3 * Co, Ci, hip.EPI_BF16, "qkv"),
                                    (B * Nq, Co, Co, hip.EPI_RESID, "proj"),
                                    (B * Nq, 4 * Co, Co, hip.EPI_GELU, "fc1"),
                                    (B * Nq, Co, 4 * Co, hip.EPI_RESID, "fc2"),
                                    (B * Nq, 4 * Co, Co, hip.EPI_DGELU, "fc2-dgrad"),
                                    (B * Nq, Co, 4 * Co, hip.EPI_F32, "fc1-dgrad"),
                                    (B * Nin, Ci, 3 * Co, hip.EPI_F32, "qkv-dgrad")]:
            if (M, N, K, epi) in seen:
                continue
            seen.add((M, N, K, epi))
            a, w = rnd(M, K), rnd(N, K)
            bias = torch.zeros(N, device=DEV)
            aux = None
            if epi == hip.EPI_RESID:
                aux = torch.zeros(M, N, device=DEV)
            if epi == hip.EPI_DGELU:
                aux = rnd(M, N)
            out = torch.empty(M, N, device=DEV, dtype=torch.float32 if epi in (hip.EPI_RESID, hip.EPI_F32) else BF16)
            out2 = torch.empty(M, N, device=DEV, dtype=BF16) if epi == hip.EPI_GELU else None
            us = timeit(lambda: ops.gemm_nt(a, w, bias, epi, out=out, out2=out2, aux=aux))
            flop = 2.0 * M * N * K
            osz = out.element_size() * M * N * (2 if epi == hip.EPI_GELU else 1)
            byts = 2 * M * K + 2 * N * K + osz + (aux.element_size() * M * N if aux is not None else 0)
            print("blk%-2d %-10s M=%6d N=%4d K=%4d  %8.1f us  %7.1f TFLOP/s  %6.2f TB/s" %
                  (blk, tag, M, N, K, us, flop / us / 1e6, byts / us / 1e6))


def bench_tn():
    print("== gemm_tn (M, N, K) ==")
    seen = set()
    for blk, Nin, Nq, Nk, Ci, Co, h, DA in BLOCKS:
        for (M, N, K, tag) in [(B * Nin, 3 * Co, Ci, "qkv-w"), (B * Nq, Co, Co, "proj-w"),
                               (B * Nq, 4 * Co, Co, "fc1-w"), (B * Nq, Co, 4 * Co, "fc2-w")]:
            if (M, N, K) in seen:
                continue
            seen.add((M, N, K))
            a, b = rnd(M, N), rnd(M, K)
            dw = torch.zeros(N, K, device=DEV)
            db = torch.zeros(N, device=DEV)
            us = timeit(lambda: ops.gemm_tn(a, b, dw, dbias=db))
            flop = 2.0 * M * N * K
            byts = 2 * M * (N + K)
            print("blk%-2d %-8s M=%6d N=%4d K=%4d  %8.1f us  %7.1f TFLOP/s  %6.2f TB/s" %
                  (blk, tag, M, N, K, us, flop / us / 1e6, byts / us / 1e6))


# 32x224^2 (BASELINE config C4, B = 4): objects 128, Nk 913 / 3265, bias columns 16 + 7 + 7 / 16 + 14 + 14
BLOCKS_C4 = [(0, 50305, 50305, 913, 96, 96, 1, 128), (1, 50305, 12673, 3265, 96, 192, 2, 160),
             (2, 12673, 12673, 913, 192, 192, 2, 128), (3, 12673, 3265, 3265, 192, 384, 4, 160),
             (4, 3265, 3265, 913, 384, 384, 4, 128), (14, 3265, 913, 3265, 384, 768, 8, 160),
             (15, 913, 913, 913, 768, 768, 8, 128)]


def bench_attn(cfg="c2"):
    """fused attention forward and backward per block shape of 16x224^2 B = 8 (c2) or 32x224^2 B = 4 (c4),
    with the bias-column counts the engine passes; step totals (blocks 4-13 counted ten times)."""
    blocks, B = (BLOCKS_C4, 4) if cfg == "c4" else (BLOCKS, 8)
    print("== attention (fwd / bwd), %s, B = %d ==" % ("32x224^2" if cfg == "c4" else "16x224^2", B))
    tot = {"f": 0.0, "b": 0.0, "alg": 0.0}
    for blk, Nin, Nq, Nk, Ci, Co, h, DA in blocks:
        if cfg == "c4":
            J = 30 if DA == 128 else 44
        else:
            J = 22 if DA == 128 else 36
        qa, ka, v = rnd(B, h, Nq, DA), (rnd(B, h, Nk, DA).float() * KSC).to(BF16), rnd(B, h, Nk, 96)
        qa[..., 96 + J:] = 0
        ka[..., 96 + J:] = 0
        scale = 96 ** -0.5
        us = timeit(lambda: ops.attn_fwd(qa, ka, v, scale, bias_cols=J))
        alg = 2.0 * B * h * Nq * Nk * 192
        ctx, lse2 = ops.attn_fwd(qa, ka, v, scale, bias_cols=J)
        dctx = rnd(B, Nq, h * 96)
        import ctypes as C
        lib = hip.load()
        lib.svit_attn_debug_set.restype, lib.svit_attn_debug_set.argtypes = C.c_int32, [C.c_int32, C.c_int32]
        lib.svit_attn_debug_set(0, 1)
        usb1 = timeit(lambda: ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, scale, bias_cols=J), iters=10)
        lib.svit_attn_debug_set(0, 2)
        usb2 = timeit(lambda: ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, scale, bias_cols=J), iters=10)
        lib.svit_attn_debug_set(0, 0)
        usb = timeit(lambda: ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, scale, bias_cols=J), iters=10)
        mult = 10 if blk == 4 else 1
        tot["f"] += us * mult; tot["b"] += usb * mult; tot["alg"] += alg * mult
        print("blk%-2d h=%d Nq=%6d Nk=%5d DA=%d J=%d  fwd %8.1f us %7.1f TF (%.3f) | bwd %8.1f us %7.1f TF (%.3f)  (dkv 4 waves %.1f, 8 waves %.1f)" %
              (blk, h, Nq, Nk, DA, J, us, alg / us / 1e6, alg / us / 1e6 / 2500, usb, 2 * alg / usb / 1e6,
               2 * alg / usb / 1e6 / 2500, usb1, usb2), flush=True)
    print("step totals (16 launches): fwd %.1f us = %.1f TF = %.3f of 2.5 PF | bwd %.1f us = %.1f TF = %.3f of 2.5 PF" %
          (tot["f"], tot["alg"] / tot["f"] / 1e6, tot["alg"] / tot["f"] / 1e6 / 2500,
           tot["b"], 2 * tot["alg"] / tot["b"] / 1e6, 2 * tot["alg"] / tot["b"] / 1e6 / 2500))


def bench_attn_fwd():
    """forward only, per block shape, with the bias-column count the engine passes (J = kt+kh+kw);
    the `attnfwd` mode of this tool."""
    print("== attention fwd ==")
    tot_us = tot_alg = 0.0
    mult = {0: 1, 1: 1, 2: 1, 3: 1, 4: 10, 14: 1, 15: 1}
    for blk, Nin, Nq, Nk, Ci, Co, h, DA in BLOCKS:
        J = 22 if DA == 128 else 36
        qa, ka, v = rnd(B, h, Nq, DA), (rnd(B, h, Nk, DA).float() * KSC).to(BF16), rnd(B, h, Nk, 96)
        qa[..., 96 + J:] = 0
        ka[..., 96 + J:] = 0
        scale = 96 ** -0.5
        us = timeit(lambda: ops.attn_fwd(qa, ka, v, scale, bias_cols=J), iters=50)
        alg = 2.0 * B * h * Nq * Nk * 192
        tot_us += us * mult[blk]
        tot_alg += alg * mult[blk]
        print("blk%-2d h=%d Nq=%6d Nk=%5d DA=%d J=%d  fwd %8.1f us %7.1f TF (%.3f of 2.5 PF)" %
              (blk, h, Nq, Nk, DA, J, us, alg / us / 1e6, alg / us / 1e6 / 2500), flush=True)
    print("step total (16 launches): %.1f us, %.1f TF = %.3f of peak" %
          (tot_us, tot_alg / tot_us / 1e6, tot_alg / tot_us / 1e6 / 2500))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "attnfwd":
    hip.load()
    bench_attn_fwd()
    sys.exit(0)

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    hip.load()
    print(torch.cuda.get_device_name(0))
    if what in ("gemm", "all"):
        bench_gemm()
    if what in ("tn", "all"):
        bench_tn()
    if what in ("attn", "all"):
        bench_attn(sys.argv[2] if len(sys.argv) > 2 else "c2")



def bench_pool():
    """pooled q/k/v kernels per block shape: fwd, ln_bwd, dgrad, wgrad (us)."""
    print("== pool kernels (which, stride) fwd / ln_bwd / dgrad / wgrad us ==")
    # (blk, heads, thw_in, stride_q, stride_kv)
    cfgs = [(0, 1, (8, 56, 56), 1, 8), (1, 2, (8, 56, 56), 2, 4), (2, 2, (8, 28, 28), 1, 4),
            (3, 4, (8, 28, 28), 2, 2), (4, 4, (8, 14, 14), 1, 2), (14, 8, (8, 14, 14), 2, 1),
            (15, 8, (8, 7, 7), 1, 1)]
    n_obj = 64
    for blk, h, thw, sq, skv in cfgs:
        N = 1 + thw[0] * thw[1] * thw[2] + n_obj
        qkv = rnd(B, N, 3, h, 96)
        w = torch.randn(96, 27, device=DEV) * 0.2
        g, b = torch.ones(96, device=DEV), torch.zeros(96, device=DEV)
        for which, s in ((0, sq), (1, skv)):
            out, pre, mean, rstd = ops.pool_ln_fwd(qkv, which, w, g, b, B, h, thw, n_obj, s)
            Nout = out.shape[2]
            t_f = timeit(lambda: ops.pool_ln_fwd(qkv, which, w, g, b, B, h, thw, n_obj, s))
            dout = rnd(B, h, Nout, 96)
            dg, db = torch.zeros(96, device=DEV), torch.zeros(96, device=DEV)
            t_l = timeit(lambda: ops.pool_ln_bwd(pre, mean, rstd, g, dg, db, B, h, Nout, d_main=dout, ld_main=96))
            dpre = ops.pool_ln_bwd(pre, mean, rstd, g, dg, db, B, h, Nout, d_main=dout, ld_main=96)
            dqkv = torch.empty_like(qkv)
            t_d = timeit(lambda: _streaming().pool_conv_dgrad(dpre, w, dqkv, which, B, h, thw, n_obj, s))
            dw = torch.zeros(96, 27, device=DEV)
            t_w = timeit(lambda: _streaming().pool_conv_wgrad(dpre, qkv, which, dw, B, h, thw, n_obj, s))
            print("blk%-2d h=%d in=%6d out=%6d which=%d s=%d  fwd %7.1f  ln_bwd %7.1f  dgrad %7.1f  wgrad %7.1f" %
                  (blk, h, N, Nout, which, s, t_f, t_l, t_d, t_w))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "pool":
    bench_pool()


def bench_tn_splits():
    print("== gemm_tn split sweep (us) ==")
    for (M, N, K) in [(13064, 1152, 384), (13064, 384, 384), (13064, 384, 1536), (50696, 576, 192),
                      (201224, 288, 96), (3656, 2304, 768)]:
        a, b = rnd(M, N), rnd(M, K)
        dw = torch.zeros(N, K, device=DEV)
        res = []
        for sp in (1, 2, 4, 8, 16, 32, 64, 0):
            us = timeit(lambda: ops.gemm_tn(a, b, dw, splits=sp), iters=10)
            res.append("%d:%.0f" % (sp, us))
        print("M=%6d N=%4d K=%4d  " % (M, N, K), "  ".join(res))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "tnsplit":
    bench_tn_splits()



def bench_nt_stages():
    """NT GEMM: pipeline depth x tile config sweep (svit_debug_set knobs)."""
    import ctypes as C
    lib = hip.load()
    lib.svit_debug_set.restype, lib.svit_debug_set.argtypes = C.c_int32, [C.c_int32, C.c_int32]
    print("== gemm_nt: us for (stages, cfg) ==")
    shapes = [(13064, 1152, 384), (13064, 1536, 384), (13064, 384, 1536), (13064, 384, 384), (50696, 192, 768),
              (13064, 384, 1152), (3656, 768, 3072), (3656, 3072, 768), (3656, 2304, 768),
              (50696, 576, 192), (50696, 192, 576), (201224, 288, 96), (201224, 96, 384)]
    for (M, N, K) in shapes:
        a, w = rnd(M, K), rnd(N, K)
        bias = torch.zeros(N, device=DEV)
        out = torch.empty(M, N, device=DEV, dtype=BF16)
        res = []
        for cfg in (0, 2, 4):
            if cfg == 0 and N % 192:
                continue
            if cfg == 4 and N % 128:
                continue
            for st in (2, 3, 4):
                for bk in (32, 64):
                    if bk == 64 and K % 64:
                        continue
                    lib.svit_debug_set(0, st)
                    lib.svit_debug_set(1, cfg)
                    lib.svit_debug_set(2, bk)
                    us = timeit(lambda: ops.gemm_nt(a, w, bias, hip.EPI_BF16, out=out), iters=10)
                    res.append("c%ds%dk%d:%.1f" % (cfg, st, bk, us))
        lib.svit_debug_set(0, 0)
        lib.svit_debug_set(1, -1)
        lib.svit_debug_set(2, 0)
        print("M=%6d N=%4d K=%4d  " % (M, N, K), " ".join(res))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ntstages":
    bench_nt_stages()


def bench_nt_ring():
    """NT GEMM: the heuristic's choice against the ring kernels (cfg 5 = 128x192, 6 = 128x96, 7 = 128x128; 2/3/4 stages),
    the pre-ring heuristic (v2) and the library's GEMM; `ntring resid|f32|gelu|dgelu` picks the epilogue (default bf16)."""
    import ctypes as C
    lib = hip.load()
    lib.svit_debug_set.restype, lib.svit_debug_set.argtypes = C.c_int32, [C.c_int32, C.c_int32]
    epi_name = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    print("== gemm_nt (%s epilogue): us, heuristic vs v2 heuristic vs ring(cfg, stages) vs torch.mm ==" % epi_name)
    shapes = [(13064, 384, 1536), (13064, 384, 1152), (3656, 768, 3072), (50696, 192, 768), (13064, 384, 2304),
              (50696, 192, 1152), (3656, 768, 2304), (50696, 192, 576), (201224, 96, 384), (201224, 96, 576),
              (13064, 384, 768), (13064, 384, 384), (13064, 1536, 384), (13064, 1152, 384), (50696, 768, 192),
              (3656, 3072, 768), (3656, 768, 768), (50696, 192, 192), (200704, 96, 448)]
    for (M, N, K) in shapes:
        a, w = rnd(M, K), rnd(N, K)
        bias = torch.zeros(N, device=DEV)
        if epi_name == "resid":
            aux, out = rnd(M, N, dtype=torch.float32), torch.empty(M, N, device=DEV)
            rs = torch.ones(8, device=DEV)
            run = lambda: ops.gemm_nt(a, w, bias, hip.EPI_RESID, out=out, aux=aux, row_scale=rs, rows_per_sample=(M + 7) // 8)
        elif epi_name == "f32":
            out = torch.empty(M, N, device=DEV)
            run = lambda: ops.gemm_nt(a, w, bias, hip.EPI_F32, out=out)
        elif epi_name == "gelu":
            out, out2 = torch.empty(M, N, device=DEV, dtype=BF16), torch.empty(M, N, device=DEV, dtype=BF16)
            run = lambda: ops.gemm_nt(a, w, bias, hip.EPI_GELU, out=out, out2=out2)
        elif epi_name == "dgelu":
            aux, out = rnd(M, N), torch.empty(M, N, device=DEV, dtype=BF16)
            run = lambda: ops.gemm_nt(a, w, None, hip.EPI_DGELU, out=out, aux=aux)
        else:
            out = torch.empty(M, N, device=DEV, dtype=BF16)
            run = lambda: ops.gemm_nt(a, w, bias, hip.EPI_BF16, out=out)
        res = []
        lib.svit_debug_set(0, 0), lib.svit_debug_set(1, -1), lib.svit_debug_set(2, 0)
        res.append("auto:%.1f" % timeit(run, iters=10))
        lib.svit_debug_set(1, 8)        # the pre-ring (v2) heuristic: what "auto" was before
        res.append("v2:%.1f" % timeit(run, iters=10))
        if K % 64 == 0:
            for cfg in (5, 6, 7):
                if (cfg == 5 and N % 192) or (cfg == 7 and N % 128):
                    continue
                for st in (2, 3, 4):
                    lib.svit_debug_set(0, st), lib.svit_debug_set(1, cfg)
                    res.append("r%ds%d:%.1f" % (cfg, st, timeit(run, iters=10)))
        lib.svit_debug_set(0, 0), lib.svit_debug_set(1, -1), lib.svit_debug_set(2, 0)
        if epi_name == "bf16":
            wt = w.t()
            res.append("lib:%.1f" % timeit(lambda: torch.mm(a, wt), iters=10))
        print("M=%6d N=%4d K=%4d  " % (M, N, K), " ".join(res), flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ntring":
    bench_nt_ring()


def bench_ln():
    """LayerNorm fwd / bwd (with the fused bf16 operand) at the residual-stream shapes (us)."""
    print("== layernorm rows x C: fwd / bwd us, bwd GB/s ==")
    for rows, C in [(8 * 25153, 96), (8 * 6337, 192), (8 * 1633, 384), (8 * 457, 768)]:
        x = rnd(rows, C, dtype=torch.float32)
        dy, dres = rnd(rows, C, dtype=torch.float32), rnd(rows, C, dtype=torch.float32)
        g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        _, _, mean, rstd = ops.layernorm_fwd(x, g, b)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        t_f = timeit(lambda: ops.layernorm_fwd(x, g, b))
        t_b = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dres=dres, want_bf16=True))
        gb = rows * C * (4 * 4 + 2) / (t_b * 1e-6) / 1e9
        print("rows %7d C %4d  fwd %7.1f  bwd %7.1f  (%.0f GB/s)" % (rows, C, t_f, t_b, gb))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ln":
    bench_ln()


def bench_tn_group():
    """grouped wgrad launch per block shape under different cost-model constants (us)."""
    import ctypes as C
    lib = hip.load()
    lib.svit_debug_set_tn.restype, lib.svit_debug_set_tn.argtypes = C.c_int32, [C.c_int32, C.c_int32]
    print("== gemm_tn_grouped: us per block group for (step_us, atomic TB/s) ==")
    groups = {}
    for blk, N_in, Nq, Nk, Ci, Co, h, DA in BLOCKS:
        M, Mq = B * N_in, B * Nq
        probs = [(Mq, Co, 4 * Co), (Mq, 4 * Co, Co), (Mq, Co, Co), (M, 3 * Co, Ci)]
        if Ci != Co:
            probs.append((M, Co, Ci))
        probs += [(B * h * Nq, 40, 96)] * 3
        groups[blk] = [(rnd(m, n), rnd(m, k), torch.zeros(n, k, device=DEV), None) for m, n, k in probs]
    lib.svit_debug_set_tn_tile.restype, lib.svit_debug_set_tn_tile.argtypes = C.c_int32, [C.c_int32]
    settings = [(85, 75, 0), (85, 75, 1), (60, 75, 1), (120, 75, 1), (85, 40, 1), (85, 130, 1), (60, 130, 1),
                (40, 130, 1), (170, 75, 1), (85, 75, 2)]
    print("blk   " + "  ".join("%3d/%3d/%d" % s for s in settings) + "   (step_us x100 / atomic TB/s x100 / tile mode)")
    tot = [0.0] * len(settings)
    for blk, g in groups.items():
        row = []
        for i, (su, at, tm) in enumerate(settings):
            lib.svit_debug_set_tn(su, at)
            lib.svit_debug_set_tn_tile(tm)
            us = timeit(lambda: ops.gemm_tn_grouped(g), iters=10)
            row.append(us)
            tot[i] += us * (10 if blk == 4 else 1)
        print("blk%-2d " % blk + "  ".join("%9.1f" % u for u in row))
    print("step  " + "  ".join("%9.0f" % u for u in tot))
    lib.svit_debug_set_tn(85, 75)
    lib.svit_debug_set_tn_tile(1)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "tngroup":
    bench_tn_group()


def bench_attn_splits():
    """attention backward: forced (dkv waves, query-split) combinations (us, whole backward)."""
    import ctypes as C
    lib = hip.load()
    lib.svit_attn_debug_set.restype, lib.svit_attn_debug_set.argtypes = C.c_int32, [C.c_int32, C.c_int32]
    print("== attn_bwd us by q_splits (0 = heuristic), 4-wave | 8-wave dkv kernel ==")
    for blk, Nin, Nq, Nk, Ci, Co, h, DA in BLOCKS:
        J = 22 if DA == 128 else 36
        qa, ka, v = rnd(B, h, Nq, DA), (rnd(B, h, Nk, DA).float() * KSC).to(BF16), rnd(B, h, Nk, 96)
        scale = 96 ** -0.5
        ctx, lse2 = ops.attn_fwd(qa, ka, v, scale, bias_cols=J)
        dctx = rnd(B, Nq, h * 96)
        res = []
        for halves in (0, 1, 2):
            lib.svit_attn_debug_set(0, halves)
            for sp in (0, 1, 2, 3, 4, 6, 8):
                if halves == 0 and sp:
                    continue
                us = timeit(lambda: ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, scale, q_splits=sp, bias_cols=J), iters=10)
                res.append("%dw/%d:%.0f" % (4 * halves, sp, us))
        lib.svit_attn_debug_set(0, 0)
        print("blk%-2d h=%d Nq=%6d Nk=%5d  " % (blk, h, Nq, Nk) + "  ".join(res), flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "attnsplits":
    bench_attn_splits()


def bench_relq():
    """query-side rel-pos forward: VALU kernel vs GEMM + gather (us)."""
    from svit_amd.engine import _rel_index
    print("== relq fwd: valu kernel | gemm + gather ==")
    cfgs = [(0, 1, (8, 56, 56), (8, 7, 7)), (2, 2, (8, 28, 28), (8, 7, 7)), (4, 4, (8, 14, 14), (8, 7, 7)),
            (14, 8, (8, 7, 7), (8, 14, 14)), (15, 8, (8, 7, 7), (8, 7, 7))]
    for blk, h, q_thw, k_thw in cfgs:
        Lq = q_thw[0] * q_thw[1] * q_thw[2]
        J = sum(k_thw)
        ld = 128 if J <= 32 else 160
        Nq = 1 + Lq + 64
        qa = rnd(B, h, Nq, ld)
        rows = [2 * max(q_thw[i], k_thw[i]) - 1 for i in (1, 2, 0)]
        tabs = [torch.randn(r, 96, device=DEV) * 0.1 for r in rows]
        idx = [_rel_index(q_thw[1], k_thw[1]).to(DEV).contiguous(), _rel_index(q_thw[2], k_thw[2]).to(DEV).contiguous(),
               _rel_index(q_thw[0], k_thw[0]).to(DEV).contiguous()]
        t_v = timeit(lambda: ops.relpos_q_fwd(qa, tabs, idx, B, h, q_thw, k_thw, 64, 96 ** 0.5))
        lp = (sum(rows) + 95) // 96 * 96
        rcat = torch.zeros(lp, 96, device=DEV, dtype=BF16)
        rcat[:sum(rows)] = torch.cat(tabs).to(BF16)
        offs = (0, rows[0], rows[0] + rows[1])

        def gg():
            P = ops.gemm_nt(qa.view(B * h * Nq, ld)[:, :96], rcat, None, hip.EPI_BF16)
            ops.relpos_gather(P, qa, idx, offs, B, h, q_thw, k_thw, 64, 96 ** 0.5)
        t_g = timeit(gg)
        print("blk%-2d h=%d Nq=%6d J=%2d  valu %7.1f | gemm+gather %7.1f" % (blk, h, Nq, J, t_v, t_g))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "relq":
    bench_relq()


def bench_pool_bwd():
    """conv dgrad + wgrad of q, k, v: two streaming launches vs the fused small-plane kernel (us)."""
    print("== pool conv backward q/k/v: dgrad3 + wgrad3 | fused ==")
    cfgs = [(4, 4, (8, 14, 14), 1, 2), (14, 8, (8, 14, 14), 2, 1), (15, 8, (8, 7, 7), 1, 1)]
    n_obj = 64
    for blk, h, thw, sq, skv in cfgs:
        N = 1 + thw[0] * thw[1] * thw[2] + n_obj
        qkv = rnd(B, N, 3, h, 96)
        ws = [torch.randn(96, 27, device=DEV) * 0.2 for _ in range(3)]
        strides = (sq, skv, skv)
        dpres = [rnd(B, h, 1 + thw[0] * ops.pooled(thw[1], s) * ops.pooled(thw[2], s) + n_obj, 96) for s in strides]
        dqkv = torch.empty_like(qkv)
        dws = [torch.zeros(96, 27, device=DEV) for _ in range(3)]
        t_d = timeit(lambda: _streaming().pool_conv_dgrad_qkv(dpres, ws, dqkv, B, h, thw, n_obj, strides))
        t_w = timeit(lambda: _streaming().pool_conv_wgrad_qkv(dpres, qkv, dws, B, h, thw, n_obj, strides))
        t_f = timeit(lambda: ops.pool_conv_bwd_qkv(dpres, ws, dqkv, qkv, dws, B, h, thw, n_obj, strides))
        print("blk%-2d h=%d N=%5d sq=%d skv=%d  dgrad3 %6.1f + wgrad3 %6.1f | fused %6.1f" % (blk, h, N, sq, skv, t_d, t_w, t_f))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "poolbwd":
    bench_pool_bwd()


def bench_pool_fwd3():
    """q/k/v pooling forward in one launch vs three (us)."""
    print("== pool_ln_fwd: three launches | fused ==")
    cfgs = [(0, 1, (8, 56, 56), 1, 8), (1, 2, (8, 56, 56), 2, 4), (2, 2, (8, 28, 28), 1, 4), (3, 4, (8, 28, 28), 2, 2),
            (4, 4, (8, 14, 14), 1, 2), (14, 8, (8, 14, 14), 2, 1), (15, 8, (8, 7, 7), 1, 1)]
    n_obj = 64
    for blk, h, thw, sq, skv in cfgs:
        N = 1 + thw[0] * thw[1] * thw[2] + n_obj
        qkv = rnd(B, N, 3, h, 96)
        ws = [torch.randn(96, 27, device=DEV) * 0.2 for _ in range(3)]
        g = [torch.ones(96, device=DEV) for _ in range(3)]
        b = [torch.zeros(96, device=DEV) for _ in range(3)]
        J = 2 * ops.pooled(thw[1], skv) + thw[0]
        da = 128 if J <= 32 else 160
        strides, lds, modes = (sq, skv, skv), (da, da, 96), (0, 1, 0)
        t3 = sum(timeit(lambda i=i: ops.pool_ln_fwd(qkv, i, ws[i], g[i], b[i], B, h, thw, n_obj, strides[i],
                                                    ld_out=lds[i], mode=modes[i])) for i in range(3))
        tf = timeit(lambda: ops.pool_ln_fwd_qkv(qkv, ws, g, b, B, h, thw, n_obj, strides, lds, modes))
        print("blk%-2d h=%d N=%6d sq=%d skv=%d  3 launches %6.1f | fused %6.1f" % (blk, h, N, sq, skv, t3, tf))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "poolfwd":
    bench_pool_fwd3()


def bench_pool_tiled():
    """q/k/v pooling forward and conv dgrad(+wgrad): streaming kernels vs the LDS-tiled stride-1
    stencils (us), per block shape of the 16x224^2 step (x multiplicity)."""
    print("== pool: streaming | tiled (stride-1 tensors) ==")
    cfgs = [(0, 1, (8, 56, 56), 1, 8, 1), (1, 2, (8, 56, 56), 2, 4, 1), (2, 2, (8, 28, 28), 1, 4, 1),
            (3, 4, (8, 28, 28), 2, 2, 1), (4, 4, (8, 14, 14), 1, 2, 10), (14, 8, (8, 14, 14), 2, 1, 1),
            (15, 8, (8, 7, 7), 1, 1, 1), (40, 4, (8, 14, 14), 1, 1, 0), (41, 2, (8, 28, 28), 1, 1, 0)]
    n_obj = 64
    tot = [0.0, 0.0, 0.0, 0.0]
    for blk, h, thw, sq, skv, mult in cfgs:
        N = 1 + thw[0] * thw[1] * thw[2] + n_obj
        qkv = rnd(B, N, 3, h, 96)
        ws = [torch.randn(96, 27, device=DEV) * 0.2 for _ in range(3)]
        g = [torch.ones(96, device=DEV) for _ in range(3)]
        b = [torch.zeros(96, device=DEV) for _ in range(3)]
        wflat = torch.cat([w.flatten() for w in ws]).contiguous()
        offs = torch.tensor([0, 2592, 5184], dtype=torch.int64, device=DEV)
        sel = ops.pool_weight_sel(wflat, offs, torch.zeros((3, 2592), dtype=torch.int32, device=DEV))
        sels = [sel[i] for i in range(3)]
        J = 2 * ops.pooled(thw[1], skv) + thw[0]
        da = 128 if J <= 32 else 160
        strides, lds, modes = (sq, skv, skv), (da, da, 96), (0, 1, 0)
        f0 = timeit(lambda: ops.pool_ln_fwd_qkv(qkv, ws, g, b, B, h, thw, n_obj, strides, lds, modes))
        f1 = timeit(lambda: ops.pool_ln_fwd_qkv(qkv, ws, g, b, B, h, thw, n_obj, strides, lds, modes, sels=sels))
        dpres = [rnd(B, h, 1 + thw[0] * ops.pooled(thw[1], s) * ops.pooled(thw[2], s) + n_obj, 96) for s in strides]
        dqkv = torch.empty_like(qkv)
        dws = [torch.zeros(96, 27, device=DEV) for _ in range(3)]
        b0 = timeit(lambda: ops.pool_conv_bwd_qkv(dpres, ws, dqkv, qkv, dws, B, h, thw, n_obj, strides))
        b1 = timeit(lambda: ops.pool_conv_bwd_qkv(dpres, ws, dqkv, qkv, dws, B, h, thw, n_obj, strides))
        for i, v in enumerate((f0, f1, b0, b1)):
            tot[i] += v * mult
        print("blk%-2d h=%d N=%6d sq=%d skv=%d  fwd %6.1f | %6.1f   conv bwd %6.1f | %6.1f" %
              (blk, h, N, sq, skv, f0, f1, b0, b1), flush=True)
    print("step totals (16 blocks): fwd %.0f | %.0f us, conv bwd %.0f | %.0f us" % tuple(tot))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "pooltiled":
    bench_pool_tiled()
