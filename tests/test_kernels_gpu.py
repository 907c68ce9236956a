"""Per-kernel parity: every C-ABI entry point of libsvit_hip.so against fp32 math on the same
inputs (the CPU oracle's functions where the op is SViT-specific).  Needs a real MI355X."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import procedural as P
from oracle import svit_ref as R

DEV = "cuda"
BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def ops():
    from svit_amd import ops as o
    from svit_amd import hip
    hip.load()
    return o


def rnd(name, shape, amp=1.0, dtype=F32):
    return P.tensor("kt:" + name, shape, amp).to(DEV).to(dtype)


def rel_err(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


def cos(got, ref):
    # (float64: an fp32 dot over millions of elements is itself only good to ~1e-4)
    got, ref = got.double().cpu().flatten(), ref.double().cpu().flatten()
    return float(torch.dot(got, ref) / (got.norm() * ref.norm() + 1e-30))


# ------------------------------------------------------------------------------ GEMMs ----
@pytest.mark.parametrize("M,K,N", [(300, 96, 96), (9000, 96, 96), (500, 384, 192), (257, 448, 96),
                                   (1000, 192, 576), (130, 768, 3072), (8200, 96, 288),
                                   # the shapes of the 14x14 / 28x28 stages, ragged M
                                   (3001, 384, 1152), (2500, 384, 384), (5000, 192, 768),
                                   (1100, 192, 576), (4099, 96, 384), (2049, 384, 288),
                                   (13064, 384, 1536)])
def test_gemm_nt_epilogues(ops, M, K, N):
    from svit_amd import hip
    a = rnd("a%d" % M, (M, K), 1.0, BF16)
    w = rnd("w%d" % N, (N, K), 0.2, BF16)
    bias = rnd("b%d" % N, (N,), 0.5)
    ref = a.float() @ w.float().t() + bias
    out = ops.gemm_nt(a, w, bias, hip.EPI_BF16)
    assert rel_err(out, ref) < 1.5e-2
    act, dact = ops.gemm_nt(a, w, bias, hip.EPI_GELU)      # gelu(h) and the saved gelu'(h)
    hr = ref.clone().requires_grad_(True)
    F.gelu(hr).backward(torch.ones_like(hr))
    assert rel_err(act, F.gelu(ref)) < 1.5e-2 and rel_err(dact, hr.grad) < 1.5e-2
    rows_per = (M + 2) // 3
    scale = torch.tensor([1.0, 0.0, 1.6667], device=DEV)
    resid = rnd("r%d" % M, (M, N), 1.0)
    out = ops.gemm_nt(a, w, bias, hip.EPI_RESID, aux=resid, row_scale=scale, rows_per_sample=rows_per)
    rs = scale[torch.arange(M, device=DEV) // rows_per][:, None]
    assert rel_err(out, resid + rs * ref) < 1e-3
    x = resid.clone()  # in place on the residual stream
    ops.gemm_nt(a, w, bias, hip.EPI_RESID, out=x, aux=x, row_scale=scale, rows_per_sample=rows_per)
    assert rel_err(x, resid + rs * ref) < 1e-3
    out = ops.gemm_nt(a, w, bias, hip.EPI_F32)
    assert rel_err(out, ref) < 1e-3
    ops.gemm_nt(a, w, None, hip.EPI_F32, out=out, accumulate=True)
    assert rel_err(out, 2 * ref - bias) < 1e-3
    dsaved = rnd("h%d" % M, (M, N), 0.6, BF16)             # fc2 dgrad: acc * saved gelu'(h)
    out = ops.gemm_nt(a, w, None, hip.EPI_DGELU, aux=dsaved)
    assert rel_err(out, (ref - bias) * dsaved.float()) < 1.5e-2


@pytest.mark.parametrize("M,N,K", [(700, 384, 128), (1333, 1152, 384), (2049, 768, 192), (515, 384, 2304),
                                   (20000, 576, 96), (130, 96, 64), (3001, 288, 448), (1301, 1536, 384)])
def test_gemm_nt_tile_variants(ops, M, N, K):
    """every (tile, pipeline depth, K-step) variant the heuristic can pick gives the same product"""
    import ctypes as C
    from svit_amd import hip
    lib = hip.load()
    lib.svit_debug_set.restype, lib.svit_debug_set.argtypes = C.c_int32, [C.c_int32, C.c_int32]
    a = rnd("va%d" % M, (M, K), 1.0, BF16)
    w = rnd("vw%d" % N, (N, K), 0.2, BF16)
    bias = rnd("vb%d" % N, (N,), 0.5)
    ref = a.float() @ w.float().t() + bias
    try:
        for cfg in (0, 2, 4):
            if (cfg == 0 and N % 192) or (cfg == 4 and N % 128):
                continue
            for st in (2, 3, 4):
                for bk in (32, 64):
                    lib.svit_debug_set(0, st), lib.svit_debug_set(1, cfg), lib.svit_debug_set(2, bk)
                    out = ops.gemm_nt(a, w, bias, hip.EPI_F32)
                    assert rel_err(out, ref) < 1e-3, (cfg, st, bk)
        # ring kernels (loader waves + MFMA waves, K-steps of 64): 128x192, 128x96, 128x128 tiles, every depth
        resid = rnd("vr%d" % M, (M, N), 1.0)
        for cfg in (5, 6, 7):
            if K % 64 or (cfg == 5 and N % 96):
                continue
            for st in (2, 3, 4):
                lib.svit_debug_set(0, st), lib.svit_debug_set(1, cfg), lib.svit_debug_set(2, 0)
                out = ops.gemm_nt(a, w, bias, hip.EPI_F32)
                assert rel_err(out, ref) < 1e-3, (cfg, st)
                out = ops.gemm_nt(a, w, bias, hip.EPI_RESID, aux=resid)
                assert rel_err(out, ref + resid) < 1e-3, (cfg, st, "resid")
                out = ops.gemm_nt(a, w, bias, hip.EPI_BF16)
                assert rel_err(out, ref) < 1e-2, (cfg, st, "bf16")
        # round 4: the one-round tiles of the wide short-K GEMMs (160x256: N % 256 == 0, 192x192: N % 192 == 0), every
        # epilogue family the 14x14 stage runs them with (bf16, GELU with both outputs, GELU-backward, fp32)
        dact = rnd("vd%d" % M, (M, N), 1.0, BF16)
        for cfg in (9, 10):
            if (cfg == 9 and N % 256) or (cfg == 10 and N % 192):
                continue
            lib.svit_debug_set(0, 0), lib.svit_debug_set(1, cfg), lib.svit_debug_set(2, 0)
            out = ops.gemm_nt(a, w, bias, hip.EPI_F32)
            assert rel_err(out, ref) < 1e-3, cfg
            out = ops.gemm_nt(a, w, bias, hip.EPI_BF16)
            assert rel_err(out, ref) < 1e-2, (cfg, "bf16")
            out2 = torch.empty(M, N, device=DEV, dtype=BF16)
            res = ops.gemm_nt(a, w, bias, hip.EPI_GELU, out2=out2)
            out = res[0] if isinstance(res, tuple) else res
            assert rel_err(out, F.gelu(ref)) < 1e-2, (cfg, "gelu")
            rg = ref.clone().requires_grad_(True)
            F.gelu(rg).sum().backward()
            assert rel_err(out2, rg.grad) < 1.5e-2, (cfg, "gelu'")
            out = ops.gemm_nt(a, w, None, hip.EPI_DGELU, aux=dact)
            assert rel_err(out, (ref - bias) * dact.float()) < 1e-2, (cfg, "dgelu")
    finally:
        lib.svit_debug_set(0, 0), lib.svit_debug_set(1, -1), lib.svit_debug_set(2, 0)


def test_gemm_nt_row_remap(ops):
    from svit_amd import hip
    B, L, N, K = 3, 50, 96, 448
    a = rnd("ra", (B * L, K), 1.0, BF16)
    w = rnd("rw", (N, K), 0.1, BF16)
    bias = rnd("rb", (N,), 0.5)
    Ntok = 1 + L + 4
    x = torch.full((B, Ntok, N), 7.0, device=DEV)
    ops.gemm_nt(a, w, bias, hip.EPI_F32, out=x, remap=(L, Ntok, 1))
    ref = (a.float() @ w.float().t() + bias).reshape(B, L, N)
    assert rel_err(x[:, 1:1 + L], ref) < 1e-3
    assert float((x[:, 0] - 7).abs().max()) == 0 and float((x[:, 1 + L:] - 7).abs().max()) == 0


@pytest.mark.parametrize("M,N,K,splits", [(1000, 288, 96, 0), (70, 96, 448, 1), (4100, 384, 1536, 0),
                                          (333, 96, 96, 3), (64, 3072, 768, 0)])
def test_gemm_tn(ops, M, N, K, splits):
    a = rnd("ta%d" % M, (M, N), 1.0, BF16)
    b = rnd("tb%d" % M, (M, K), 1.0, BF16)
    dw = torch.ones((N, K), device=DEV)
    db = torch.ones(N, device=DEV)
    ops.gemm_tn(a, b, dw, splits, dbias=db)
    ref = a.float().t() @ b.float() + 1.0
    assert rel_err(dw, ref) < 2e-4
    assert rel_err(db, a.float().sum(0) + 1.0) < 2e-4


@pytest.mark.parametrize("tile_mode", [1, 0, 2])
@pytest.mark.parametrize("count", [1, 5, 11])
def test_gemm_tn_grouped(ops, count, tile_mode):
    """several wgrads per launch (incl. strided column slices, K tail inside a padded ldb, no
    bias) == the same GEMMs one by one; tile_mode: 1 = the heuristic (128x192 tiles where K is a
    multiple of 192), 0 = 128x96 everywhere, 2 = 128x192 everywhere (partial tiles masked)"""
    import ctypes as C
    from svit_amd import hip
    lib = hip.load()
    lib.svit_debug_set_tn_tile.restype, lib.svit_debug_set_tn_tile.argtypes = C.c_int32, [C.c_int32]
    lib.svit_debug_set_tn_tile(tile_mode)
    try:
        _tn_grouped_case(ops, count)
    finally:
        lib.svit_debug_reset()


def _tn_grouped_case(ops, count, shapes=None):
    shapes = (shapes or [(1000, 288, 96), (4100, 384, 1536), (70, 96, 441), (13064, 384, 384), (333, 40, 96),
              (64, 3072, 768), (2000, 1152, 384), (5000, 96, 96), (129, 128, 96), (8000, 64, 96),
              (700, 768, 768)])[:count]
    probs, refs = [], []
    for i, (M, N, K) in enumerate(shapes):
        wide = rnd("ga%d" % i, (M, N + 16), 1.0, BF16)
        a = wide[:, 8:8 + N]                                  # row-strided view
        b = rnd("gb%d" % i, (M, (K + 7) // 8 * 8), 1.0, BF16)
        dw = torch.full((N, K), 0.5, device=DEV)
        db = torch.full((N,), 0.25, device=DEV) if i % 3 != 2 else None
        probs.append((a, b, dw, db))
        refs.append((a.float().t() @ b.float()[:, :K] + 0.5,
                     None if db is None else a.float().sum(0) + 0.25))
    ops.gemm_tn_grouped(probs)
    for (a, b, dw, db), (rw, rb) in zip(probs, refs):
        assert rel_err(dw, rw) < 2e-4
        if db is not None:
            assert rel_err(db, rb) < 2e-4


def test_colsum_cast_scale(ops):
    a = rnd("cs", (1234, 288), 1.0, BF16)
    out = torch.zeros(288, device=DEV)
    ops.colsum(a, out)
    assert rel_err(out, a.float().sum(0)) < 1e-4
    src = rnd("cast", (1000003,), 3.0)
    assert torch.equal(ops.cast_bf16(src), src.to(BF16))
    x = rnd("sc", (6, 50, 96), 1.0)
    s = torch.tensor([1.0, 0.0, 2.5], device=DEV)
    got = ops.scale_cast(x.reshape(300, 96), s, 100)
    ref = (x.reshape(3, 100, 96) * s[:, None, None]).to(BF16).reshape(300, 96)
    assert torch.equal(got, ref)
    assert torch.equal(ops.scale_cast(x.reshape(300, 96)), x.reshape(300, 96).to(BF16))
    got = ops.scale_cast(x, gather=(40, 3))                # rows [3, 43) of each of the 6 samples
    assert torch.equal(got, x[:, 3:43].reshape(240, 96).to(BF16))


def test_transpose_cast_batched(ops):
    mats = [(96, 288), (100, 37), (768, 3072)]
    total = sum(r * c for r, c in mats)
    src = rnd("tp", (total,), 1.0)
    dst = torch.zeros(total, device=DEV, dtype=BF16)
    table, off = [], 0
    for r, c in mats:
        table += [off, off, r, c, r]
        off += r * c
    tab = torch.tensor(table, dtype=torch.int64, device=DEV)
    ops.transpose_cast_batched(src, dst, tab, len(mats), 256)
    off = 0
    for r, c in mats:
        ref = src[off:off + r * c].reshape(r, c).t().contiguous().to(BF16)
        assert torch.equal(dst[off:off + r * c].reshape(c, r), ref)
        off += r * c


def test_transpose_bf16_batched(ops):
    """the step's form: from the bf16 mirror; odd shapes, a table placed inside a wider row (ldd > R), unaligned offsets"""
    mats = [(96, 288, 96), (100, 37, 104), (768, 3072, 768), (27, 96, 64), (1536, 384, 1536)]
    src_off, dst_off, so, do = [], [], 0, 3          # (dst base off by 3 elements: the scalar store path)
    for r, c, ldd in mats:
        src_off.append(so); dst_off.append(do)
        so += (r * c + 7) // 8 * 8
        do += c * ldd + 5
    src = rnd("tpb", (so,), 1.0).to(BF16)
    dst = torch.full((do + 8,), 7.0, device=DEV, dtype=BF16)
    table = []
    for (r, c, ldd), a, b in zip(mats, src_off, dst_off):
        table += [a, b, r, c, ldd]
    tab = torch.tensor(table, dtype=torch.int64, device=DEV)
    ops.transpose_bf16_batched(src, dst, tab, len(mats), 256)
    for (r, c, ldd), a, b in zip(mats, src_off, dst_off):
        got = dst[b:b + c * ldd].reshape(c, ldd)
        assert torch.equal(got[:, :r], src[a:a + r * c].reshape(r, c).t())
        if ldd > r:
            assert bool((got[:-1, r:] == 7.0).all())      # columns past R are not touched
    # aligned destination (the product's layout): the 16-byte store path
    dst2 = torch.zeros(so, device=DEV, dtype=BF16)
    tab2 = torch.tensor(sum([[a, a, r, c, r] for (r, c, _), a in zip(mats, src_off)], []), dtype=torch.int64, device=DEV)
    ops.transpose_bf16_batched(src, dst2, tab2, len(mats), 256)
    for (r, c, _), a in zip(mats, src_off):
        assert torch.equal(dst2[a:a + r * c].reshape(c, r), src[a:a + r * c].reshape(r, c).t())


# -------------------------------------------------------------------------- LayerNorm ----
@pytest.mark.parametrize("C", [96, 192, 384, 768])
def test_layernorm(ops, C):
    rows = 1037
    x = rnd("lnx%d" % C, (rows, C), 2.0) + 0.3
    g = rnd("lng%d" % C, (C,), 0.2) + 1.0
    b = rnd("lnb%d" % C, (C,), 0.1)
    y16, y32, mean, rstd = ops.layernorm_fwd(x, g, b, want_f32=True)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-6)
    assert rel_err(y32, ref) < 1e-5 and rel_err(y16, ref) < 1e-2
    dy = rnd("lnd%d" % C, (rows, C), 1.0)
    dres = rnd("lnr%d" % C, (rows, C), 1.0)
    ref.backward(dy)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx = ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dres=dres)
    assert rel_err(dx, xr.grad + dres) < 1e-4
    assert rel_err(dg, gr.grad) < 1e-4 and rel_err(db, br.grad) < 1e-4
    # fused bf16 operand for the next GEMM: bf16(DropPath scale[row / rows_per_sample] * dx)
    dg2, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    s = torch.tensor([0.0, 1.25, 2.0, 1.0, 0.5], device=DEV)       # 5 samples of 208 rows (last short)
    dx2, dx16 = ops.layernorm_bwd(dy, x, g, mean, rstd, dg2, db2, dres=dres, want_bf16=True,
                                  row_scale=s, rows_per_sample=208)
    assert torch.equal(dx2, dx)
    exp = (dx * s[torch.arange(rows, device=DEV) // 208][:, None]).to(BF16)
    assert torch.equal(dx16, exp)
    _, dx16b = ops.layernorm_bwd(dy, x, g, mean, rstd, dg2, db2, dres=dres, want_bf16=True)
    assert torch.equal(dx16b, dx.to(BF16))
    # bf16 upstream gradient (what the dgrad GEMM writes): same result as its f32 widening
    dg3, db3 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dg4, db4 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dxa = ops.layernorm_bwd(dy.to(BF16), x, g, mean, rstd, dg3, db3, dres=dres)
    dxb = ops.layernorm_bwd(dy.to(BF16).float(), x, g, mean, rstd, dg4, db4, dres=dres)
    assert torch.equal(dxa, dxb) and rel_err(dg3, dg4) < 1e-6 and rel_err(db3, db4) < 1e-6


# -------------------------------------------------------------------- patch embedding ----
@pytest.mark.parametrize("T,H,W", [(4, 36, 36), (2, 24, 312), (3, 30, 39)])
def test_patch_embed(ops, T, H, W):
    """square crop, a row wider than one LDS chunk (BASELINE 312^2 crop) and an odd width"""
    from svit_amd import hip
    B = 2
    video = rnd("vid%d" % W, (B, 3, T, H, W), 1.7)
    w = rnd("pw", (96, 3, 3, 7, 7), 0.08)
    bias = rnd("pb", (96,), 0.05)
    cols, (To, Ho, Wo) = ops.im2col_patch(video)
    ref_cols = F.conv3d  # silence linters
    wk = torch.zeros((96, 448), device=DEV)
    wk[:, :441] = w.reshape(96, 441)
    out = ops.gemm_nt(cols, wk.to(BF16), bias, hip.EPI_F32)
    ref = F.conv3d(video, w, bias, stride=(2, 4, 4), padding=(1, 3, 3))
    assert ref.shape[2:] == (To, Ho, Wo)
    ref = ref.flatten(2).transpose(1, 2).reshape(-1, 96)
    assert rel_err(out, ref) < 1.5e-2 and cos(out, ref) > 0.9999


def test_special_tokens(ops):
    B, L, Tx, O, C = 2, 10, 4, 4, 96
    N = 1 + L + Tx * O
    x = torch.zeros((B, N, C), device=DEV)
    cls, objq, pos = rnd("cls", (C,)), rnd("oq", (O, C)), rnd("pt", (Tx, C))
    ops.fill_special_tokens(x, cls, objq, pos, L, Tx, O, True)
    assert torch.equal(x[:, 0], cls.expand(B, C))
    ref = (objq[None, :, :] + pos[:, None, :]).reshape(Tx * O, C)
    assert torch.equal(x[:, 1 + L:], ref.expand(B, -1, -1))
    assert float(x[:, 1:1 + L].abs().max()) == 0


# ------------------------------------------------------------------ pooled q/k/v path ----
def _qkv(B, h, thw, O, tag):
    N = 1 + thw[0] * thw[1] * thw[2] + O
    return rnd("qkv:" + tag, (B, N, 3, h, 96), 1.0, BF16)


@pytest.mark.parametrize("stride,thw", [(1, (2, 8, 8)), (2, (2, 8, 8)), (2, (2, 7, 7)), (4, (2, 8, 8)),
                                        (8, (3, 16, 16)), (2, (1, 9, 9))])
def test_pool_ln_fwd_bwd(ops, stride, thw):
    B, h, O = 2, 2, 3
    qkv = _qkv(B, h, thw, O, "p%d" % stride)
    w = rnd("pcw%d" % stride, (96, 1, 3, 3, 3), 0.3)
    g = rnd("pg", (96,), 0.2) + 1.0
    b = rnd("pb2", (96,), 0.1)
    which = 1
    Ho, Wo = ops.pooled(thw[1], stride), ops.pooled(thw[2], stride)
    ld = 128 if (Ho + Wo + thw[0]) <= 32 else 160
    out, pre, mean, rstd = ops.pool_ln_fwd(qkv, which, w.reshape(96, 27).contiguous(), g, b, B, h,
                                           thw, O, stride, ld_out=ld, mode=1)
    x = qkv[:, :, which].permute(0, 2, 1, 3).float().cpu().requires_grad_(True)   # [B,h,N,96]
    wc = w.cpu().requires_grad_(True)
    gc, bc = g.cpu().requires_grad_(True), b.cpu().requires_grad_(True)
    ref, thw_o = R.pool_tokens(x, thw, (1, stride, stride), wc, gc, bc, O)
    assert thw_o == (thw[0], Ho, Wo)
    assert rel_err(out[..., :96], ref) < 2e-2 and cos(out[..., :96], ref) > 0.9999
    # one-hot key coordinates
    oh = out[..., 96:].float().cpu()
    Lo = thw[0] * Ho * Wo
    assert float(oh[:, :, 0].abs().max()) == 0 and float(oh[:, :, 1 + Lo:].abs().max()) == 0
    p = torch.arange(Lo)
    exp = torch.zeros(Lo, ld - 96)
    exp[p, (p // Wo) % Ho] = 1
    exp[p, Ho + p % Wo] = 1
    exp[p, Ho + Wo + p // (Wo * Ho)] = 1
    assert torch.equal(oh[0, 0, 1:1 + Lo], exp) and torch.equal(oh[1, 1, 1:1 + Lo], exp)
    # backward: LN bwd + conv dgrad + conv wgrad (+ object gain) against autograd of the oracle
    Nout = ref.shape[2]
    dout = rnd("pd%d" % stride, (B, h, Nout, 96), 1.0, BF16)
    ref.backward(dout.float().cpu())
    dgam, dbet = torch.zeros(96, device=DEV), torch.zeros(96, device=DEV)
    dpre = ops.pool_ln_bwd(pre, mean, rstd, g, dgam, dbet, B, h, Nout, d_main=dout, ld_main=96)
    assert rel_err(dgam, gc.grad) < 2e-2 and rel_err(dbet, bc.grad) < 2e-2
    # conv dgrad + wgrad: the fused q-k-v entry point (the product's only conv backward since round 6) with this tensor's
    # dpre / weight / stride in all three slots -- slice `which` of dqkv and its dw are what autograd of the oracle gives
    dqkv = torch.full_like(qkv, 7.0)
    w27 = w.reshape(96, 27).contiguous()
    dws = [torch.zeros((96, 27), device=DEV) for _ in range(3)]
    ops.pool_conv_bwd_qkv([dpre] * 3, [w27] * 3, dqkv, qkv, dws, B, h, thw, O, (stride,) * 3)
    from svit_amd import hip
    assert hip.load().svit_debug_pool_bwd_path() == 1
    got = dqkv[:, :, which].permute(0, 2, 1, 3)
    assert cos(got, x.grad) > 0.9995 and rel_err(got, x.grad) < 3e-2
    assert torch.equal(dqkv[:, :, 0], dqkv[:, :, which])         # (dx depends on dpre and w only)
    assert cos(dws[which], wc.grad.reshape(96, 27)) > 0.9995 and rel_err(dws[which], wc.grad.reshape(96, 27)) < 3e-2


@pytest.mark.parametrize("sq,skv,thw", [(1, 1, (2, 8, 8)), (1, 2, (3, 7, 7)), (2, 1, (2, 16, 16)),
                                         (1, 8, (2, 30, 30)), (1, 1, (1, 14, 14)), (1, 1, (5, 5, 9)),
                                         (2, 2, (4, 14, 14)), (1, 2, (8, 14, 14)), (2, 2, (2, 28, 28)),
                                         (1, 1, (8, 7, 7)), (1, 2, (3, 56, 56)), (2, 4, (2, 33, 31))])
def test_pool_tiled_stride1_equals_streaming(ops, sq, skv, thw):
    """LDS-tiled / slab stride-1 stencils (whichever the planner picks for the plane: halo ring in LDS
    with the LayerNorm spanning four waves, or the slab kernel with scalar weights; and the conv
    dgrad with the flipped kernel) against the streaming kernels they replace.  Same taps, but the
    accumulation order and -ffast-math contraction differ between the code paths, so the claim is
    "equal to one bf16 ulp" (rel_err < 2e-2, cosine > 0.9999), NOT bit identity; only the one-hot
    rel-pos columns, which are copied, must be exactly equal.  Covers partial tiles, the 7x7 tile,
    two x tiles, a t walk cut in chunks, T = 1 and non-square planes."""
    B, h, O = 2, 2, 3
    qkv = _qkv(B, h, thw, O, "t%d%d%d" % (sq, skv, thw[1]))
    ws = [rnd("tw%d" % i, (96, 27), 0.3) for i in range(3)]
    gs = [rnd("tg%d" % i, (96,), 0.2) + 1.0 for i in range(3)]
    bs = [rnd("tb%d" % i, (96,), 0.1) for i in range(3)]
    strides, lds, modes = (sq, skv, skv), (128, 128, 96), (0, 1, 0)
    if thw[0] + ops.pooled(thw[1], skv) + ops.pooled(thw[2], skv) > 32:
        lds = (160, 160, 96)
    wflat = torch.cat([w.flatten() for w in ws]).contiguous()
    offs = torch.tensor([0, 96 * 27, 2 * 96 * 27], dtype=torch.int64, device=DEV)
    sel = ops.pool_weight_sel(wflat, offs, torch.zeros((3, 27 * 96), dtype=torch.int32, device=DEV))
    sels = [sel[i] for i in range(3)]
    ref = ops.pool_ln_fwd_qkv(qkv, ws, gs, bs, B, h, thw, O, strides, lds, modes)
    got = ops.pool_ln_fwd_qkv(qkv, ws, gs, bs, B, h, thw, O, strides, lds, modes, sels=sels)
    for i in range(3):
        cols = slice(0, 96) if modes[i] == 0 else slice(None)
        # same taps in the same order; -ffast-math may still contract the two code paths
        # differently, so: equal to one bf16 ulp, and (mode 1) the one-hot columns exactly
        assert rel_err(got[i][0][..., :96], ref[i][0][..., :96]) < 2e-2 and cos(got[i][0][..., :96], ref[i][0][..., :96]) > 0.9999, i
        assert torch.equal(got[i][0][..., 96:][..., cols if modes[i] else slice(0, 0)], ref[i][0][..., 96:][..., cols if modes[i] else slice(0, 0)]), i
        assert rel_err(got[i][1], ref[i][1]) < 2e-2 and cos(got[i][1], ref[i][1]) > 0.9999, i
        assert rel_err(got[i][2], ref[i][2]) < 2e-4 and rel_err(got[i][3], ref[i][3]) < 2e-4, i   # (another conv summation order on the slab planes since round 4)


@pytest.mark.parametrize("sq,skv,thw", [(1, 2, (2, 8, 8)), (2, 1, (3, 7, 7)), (1, 8, (2, 16, 16)),
                                         (2, 4, (2, 12, 12))])
def test_pool_qkv_fused_equals_single(ops, sq, skv, thw):
    """the one-launch q/k/v entry points (what the engine calls) against the per-tensor ones"""
    B, h, O = 2, 2, 3
    qkv = _qkv(B, h, thw, O, "f%d%d" % (sq, skv))
    ws = [rnd("fw%d" % i, (96, 27), 0.3) for i in range(3)]
    gs = [rnd("fg%d" % i, (96,), 0.2) + 1.0 for i in range(3)]
    bs = [rnd("fb%d" % i, (96,), 0.1) for i in range(3)]
    strides, lds, modes = (sq, skv, skv), (128, 128, 96), (0, 1, 0)
    fused = ops.pool_ln_fwd_qkv(qkv, ws, gs, bs, B, h, thw, O, strides, lds, modes)
    single = [ops.pool_ln_fwd(qkv, i, ws[i], gs[i], bs[i], B, h, thw, O, strides[i], ld_out=lds[i],
                              mode=modes[i]) for i in range(3)]
    for i, (fu, si) in enumerate(zip(fused, single)):
        cols = slice(0, 96) if modes[i] == 0 else slice(None)     # q's extra columns: written later
        assert torch.equal(fu[0][..., cols], si[0][..., cols])
        assert torch.equal(fu[1], si[1]) and torch.equal(fu[2], si[2]) and torch.equal(fu[3], si[3])
    entries, refs = [], []
    dgs = [torch.zeros(96, device=DEV) for _ in range(6)]
    dbs = [torch.zeros(96, device=DEV) for _ in range(6)]
    for i in range(3):
        out, pre, mean, rstd = single[i]
        Nout = out.shape[2]
        dout = rnd("fd%d%d" % (i, sq), (B, h, Nout, 96), 1.0, BF16)
        entries.append(((pre, mean, rstd, gs[i], dgs[i], dbs[i], B, h, Nout), dict(d_main=dout, ld_main=96)))
        refs.append(ops.pool_ln_bwd(pre, mean, rstd, gs[i], dgs[3 + i], dbs[3 + i], B, h, Nout,
                                    d_main=dout, ld_main=96))
    dpres = ops.pool_ln_bwd_qkv(entries)
    for i in range(3):
        assert torch.equal(dpres[i], refs[i])
        assert rel_err(dgs[i], dgs[3 + i]) < 1e-5 and rel_err(dbs[i], dbs[3 + i]) < 1e-5


@pytest.mark.parametrize("sq,skv,thw,B,h", [(1, 2, (4, 14, 14), 2, 4), (2, 1, (2, 14, 14), 2, 8),
                                             (1, 1, (3, 7, 7), 3, 2), (2, 2, (2, 9, 11), 2, 2),
                                             (1, 4, (2, 16, 16), 2, 2),
                                             (1, 2, (8, 14, 14), 2, 4),      # blocks 4-13 of 16x224^2 (chunked in t)
                                             (2, 2, (8, 28, 28), 1, 4),      # block 3: 28x28 planes pooled to 14x14
                                             (2, 1, (8, 14, 14), 1, 8),      # block 14
                                             (1, 1, (8, 7, 7), 2, 8),        # block 15
                                             (1, 2, (16, 14, 14), 1, 4),     # 32x224^2
                                             (1, 2, (1, 14, 14), 3, 4),      # frames pass / image ranks (T' = 1)
                                             (1, 2, (5, 13, 9), 2, 1),       # odd planes, odd T
                                             (2, 2, (3, 20, 20), 1, 2),      # 312^2 crop, block 4 (20x20 -> 10x10)
                                             (1, 8, (2, 56, 56), 1, 1),      # block 0: 56x56 planes cut in y, k / v at stride 8
                                             (2, 4, (3, 56, 56), 1, 2),      # block 1: 56x56 -> 28x28 (q), stride 4 (k, v)
                                             (1, 4, (3, 28, 28), 2, 2),      # block 2
                                             (1, 8, (2, 30, 23), 1, 1),      # ragged windows at stride 8 (the last one runs to the edge)
                                             (2, 3, (2, 19, 26), 1, 2)])     # stride 3
def test_pool_conv_bwd_fused_small_planes(ops, sq, skv, thw, B, h):
    """Round 5: conv dgrad + conv wgrad of q, k, v in ONE launch with the dpre halo staged once in LDS
    (csrc/pool.hip::pool_bwd_fused_kernel; every stride, planes cut in t and y).  Round 6: against fp32 autograd of the
    pre-LayerNorm pooling on the CPU (depthwise Conv3d + the object-gain rows + the cls row: attention.py:13-65 as
    oracle.svit_ref.pool_tokens restates it, without its LayerNorm) on the same bf16 dpre -- the streaming kernels this test
    used as its comparator in round 5 left the product library."""
    from svit_amd import hip
    lib = hip.load()
    O = 5
    qkv = _qkv(B, h, thw, O, "s%d%d" % (sq, skv))
    ws = [rnd("sw%d" % i, (96, 27), 0.3) for i in range(3)]
    strides = (sq, skv, skv)
    T, H, W = thw
    L = T * H * W
    dpres, refs = [], []
    for i, s_ in enumerate(strides):
        Nout = 1 + T * ops.pooled(H, s_) * ops.pooled(W, s_) + O
        dpres.append(rnd("sd%d%d" % (i, s_), (B, h, Nout, 96), 1.0, BF16))
        x = qkv[:, :, i].permute(0, 2, 1, 3).float().cpu().requires_grad_(True)       # [B, h, N, 96]
        wc = ws[i].cpu().reshape(96, 1, 3, 3, 3).clone().requires_grad_(True)
        vol = x[:, :, 1:1 + L].reshape(B * h, T, H, W, 96).permute(0, 4, 1, 2, 3)
        vol = F.conv3d(vol, wc, None, stride=(1, s_, s_), padding=1, groups=96)
        pre = torch.cat([x[:, :, :1], vol.reshape(B, h, 96, -1).transpose(2, 3),
                         x[:, :, 1 + L:] * R.object_gain(wc, (1, s_, s_))], dim=2)
        pre.backward(dpres[i].float().cpu())
        refs.append((x.grad, wc.grad.reshape(96, 27)))
    d_got = torch.full_like(qkv, 7.0)                       # every element must be overwritten
    dw_got = [torch.full((96, 27), 0.5, device=DEV) for _ in range(3)]
    ops.pool_conv_bwd_qkv(dpres, ws, d_got, qkv, dw_got, B, h, thw, O, strides)
    assert lib.svit_debug_pool_bwd_path() == 1, "the fused kernel did not run"
    for i in range(3):
        gx, gw = refs[i]
        got = d_got[:, :, i].permute(0, 2, 1, 3)
        assert cos(got, gx) > 0.9999 and rel_err(got, gx) < 1.5e-2, i      # bf16 weights in the dgrad taps, bf16 output
        assert rel_err(dw_got[i] - 0.5, gw) < 2e-3 and cos(dw_got[i] - 0.5, gw) > 0.99999, i
    # the product library has no other conv backward: with the fused kernel refused the entry point fails loudly
    try:
        assert lib.svit_debug_set_pool(1, 0) == 0
        with pytest.raises(hip.SvitHipError):
            ops.pool_conv_bwd_qkv(dpres, ws, torch.empty_like(qkv), qkv, [torch.zeros((96, 27), device=DEV) for _ in range(3)],
                                  B, h, thw, O, strides)
        assert lib.svit_debug_pool_bwd_path() == 0
    finally:
        lib.svit_debug_reset()


@pytest.mark.parametrize("sq,skv,thw,h", [(1, 8, (2, 56, 56), 1),       # block 0 (two of its eight planes)
                                           (2, 4, (2, 56, 56), 2),       # block 1
                                           (1, 4, (3, 28, 28), 2),       # block 2
                                           (2, 2, (8, 28, 28), 1),       # block 3
                                           (1, 2, (8, 14, 14), 2),       # blocks 4-13
                                           (2, 1, (8, 14, 14), 2),       # block 14
                                           (1, 1, (8, 7, 7), 2),         # block 15
                                           (1, 2, (1, 14, 14), 2)])      # T' = 1
def test_pool_backward_vs_oracle_at_the_step_shapes(ops, sq, skv, thw, h):
    """The pooling backward as the engine runs it at every (stride, plane) of blocks 3-15 -- svit_pool_ln_bwd_qkv then
    the fused conv backward (blocks 0-15: every stride of the model) -- against autograd of the oracle's attention_pool restatement (oracle.svit_ref.pool_tokens,
    attention.py:13-65): d(qkv), d(conv weight), d(gamma), d(beta) of q, k and v."""
    B, O = 1, 8
    qkv = _qkv(B, h, thw, O, "o%d%d%d" % (sq, skv, thw[1]))
    strides = (sq, skv, skv)
    ws = [rnd("ow%d" % i, (96, 27), 0.3) for i in range(3)]
    gs = [rnd("og%d" % i, (96,), 0.2) + 1.0 for i in range(3)]
    bs = [rnd("ob%d" % i, (96,), 0.1) for i in range(3)]
    entries, refs = [], []
    for i, s in enumerate(strides):
        Ho, Wo = ops.pooled(thw[1], s), ops.pooled(thw[2], s)
        out, pre, mean, rstd = ops.pool_ln_fwd(qkv, i, ws[i], gs[i], bs[i], B, h, thw, O, s, ld_out=96, mode=0)
        x = qkv[:, :, i].permute(0, 2, 1, 3).float().cpu().requires_grad_(True)
        wc = ws[i].cpu().reshape(96, 1, 3, 3, 3).clone().requires_grad_(True)
        gc, bc = gs[i].cpu().clone().requires_grad_(True), bs[i].cpu().clone().requires_grad_(True)
        ref, _ = R.pool_tokens(x, thw, (1, s, s), wc, gc, bc, O)
        assert rel_err(out, ref) < 2e-2
        Nout = ref.shape[2]
        dout = rnd("od%d%d" % (i, s), (B, h, Nout, 96), 1.0, BF16)
        ref.backward(dout.float().cpu())
        refs.append((x.grad, wc.grad.reshape(96, 27), gc.grad, bc.grad))
        dg, db = torch.zeros(96, device=DEV), torch.zeros(96, device=DEV)
        entries.append(((pre, mean, rstd, gs[i], dg, db, B, h, Nout), dict(d_main=dout, ld_main=96)))
    dpres = ops.pool_ln_bwd_qkv(entries)
    dqkv = torch.full_like(qkv, 7.0)
    dws = [torch.zeros((96, 27), device=DEV) for _ in range(3)]
    ops.pool_conv_bwd_qkv(dpres, ws, dqkv, qkv, dws, B, h, thw, O, strides)
    from svit_amd import hip
    assert hip.load().svit_debug_pool_bwd_path() == 1, "the step's shapes must take the fused kernel"
    for i in range(3):
        gx, gw, gg, gb = refs[i]
        got = dqkv[:, :, i].permute(0, 2, 1, 3)
        assert cos(got, gx) > 0.9995 and rel_err(got, gx) < 3e-2, i
        assert cos(dws[i], gw) > 0.9995 and rel_err(dws[i], gw) < 3e-2, i
        assert rel_err(entries[i][0][4], gg) < 2e-2 and rel_err(entries[i][0][5], gb) < 2e-2, i


def test_pool_ln_bwd_three_inputs(ops):
    B, h, Nout = 2, 2, 37
    pre = rnd("pre3", (B, h, Nout, 96), 1.0, BF16)
    g = rnd("g3", (96,), 0.2) + 1.0
    x = pre.float()
    mean = x.mean(-1).flatten().contiguous()
    rstd = (1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-6)).flatten().contiguous()
    d_main = rnd("dm3", (B, h, Nout, 128), 1.0, BF16)
    d_res = rnd("dr3", (B, Nout, h * 96), 1.0, BF16)
    d_extra = rnd("de3", (B, h, Nout, 96), 1.0)
    dg, db = torch.zeros(96, device=DEV), torch.zeros(96, device=DEV)
    dpre = ops.pool_ln_bwd(pre, mean, rstd, g, dg, db, B, h, Nout, d_main=d_main, ld_main=128,
                           d_res=d_res, d_extra=d_extra)
    res = d_res.float().reshape(B, Nout, h, 96).permute(0, 2, 1, 3).clone()
    res[:, :, 0] = 0  # the residual-pooling path skips cls
    dsum = d_main[..., :96].float() + res + d_extra
    xr = x.clone().requires_grad_(True)
    F.layer_norm(xr, (96,), g, torch.zeros_like(g), 1e-6).backward(dsum)
    assert rel_err(dpre, xr.grad) < 2e-2 and cos(dpre, xr.grad) > 0.9999


@pytest.mark.parametrize("B,h,thw,sq,skv,n_obj,slab", [
    (2, 4, (8, 14, 14), 1, 2, 64, True),       # blocks 4-13: the slab LayerNorm kernel multiplies q . R^T itself
    (2, 8, (8, 14, 14), 2, 1, 64, True),       # block 14 (q pooled to 7x7, 36 key coordinates: DA = 160)
    (1, 4, (8, 28, 28), 2, 2, 64, False),      # 28x28: not a slab plane -> staged conv + the same LayerNorm kernel (round 5)
    (1, 1, (2, 56, 56), 1, 8, 8, False),       # block 0: 240 table rows (Lpad 288), the widest product tile the kernel takes
    (2, 4, (8, 14, 14), 1, 2, 64, None),       # no selector tables at all (streaming kernels) -> GEMM launch
])
def test_pool_qkv_writes_the_relpos_columns(ops, B, h, thw, sq, skv, n_obj, slab):
    """svit_pool_args.relq_*: the q tensor's rel-pos columns qa[..., 96 + j] come out of the pooling call -- from
    the slab LayerNorm kernel's own MFMA product where q takes that path, from an SVIT_EPI_RELQ GEMM the entry
    point launches otherwise -- BIT-identical to the stand-alone GEMM on the qa the same call produced, and the
    other outputs unchanged."""
    from svit_amd import hip
    T, H, W = thw
    L = T * H * W
    N = 1 + L + n_obj
    q_thw = (T, ops.pooled(H, sq), ops.pooled(W, sq))
    k_thw = (T, ops.pooled(H, skv), ops.pooled(W, skv))
    kt, kh, kw = k_thw
    J = kt + kh + kw
    da = 128 if J <= 32 else 160
    Lq = q_thw[0] * q_thw[1] * q_thw[2]
    Nq = 1 + Lq + n_obj
    qkv = rnd("rq%d%d%d" % (T, H, h), (B, N, 3, h, 96), 0.5, BF16)
    ws = [rnd("rw%d%d" % (i, H), (96, 27), 0.2) for i in range(3)]
    g = [rnd("rg%d" % i, (96,), 0.3) + 1.0 for i in range(3)]
    b = [rnd("rb%d" % i, (96,), 0.1) for i in range(3)]
    sels = None
    if slab is not None:
        wflat = torch.cat([w.flatten() for w in ws]).contiguous()
        offs = torch.tensor([0, 2592, 5184], dtype=torch.int64, device=DEV)
        sel = ops.pool_weight_sel(wflat, offs, torch.zeros((3, 2592), dtype=torch.int32, device=DEV))
        sels = [sel[i] for i in range(3)]
    rows = [2 * max(q_thw[i], k_thw[i]) - 1 for i in (1, 2, 0)]
    rows_off = (0, rows[0], rows[0] + rows[1])
    lp = (sum(rows) + 95) // 96 * 96
    rcat = torch.zeros((lp, 96), device=DEV, dtype=BF16)
    rcat[:sum(rows)] = rnd("rr%d%d" % (H, h), (sum(rows), 96), 0.3, BF16)
    idx = [R.rel_index(q_thw[1], k_thw[1]), R.rel_index(q_thw[2], k_thw[2]), R.rel_index(q_thw[0], k_thw[0])]
    body = torch.full((q_thw[0], q_thw[1], q_thw[2], da - 96), -1, dtype=torch.int32)
    body[..., :kh] = (rows_off[0] + idx[0].to(torch.int32)).view(1, q_thw[1], 1, kh)
    body[..., kh:kh + kw] = (rows_off[1] + idx[1].to(torch.int32)).view(1, 1, q_thw[2], kw)
    body[..., kh + kw:J] = (rows_off[2] + idx[2].to(torch.int32)).view(q_thw[0], 1, 1, kt)
    cmap = torch.full((Nq, da - 96), -1, dtype=torch.int32)
    cmap[1:1 + Lq] = body.view(Lq, da - 96)
    cmap = cmap.to(DEV).contiguous()
    args = (qkv, ws, g, b, B, h, thw, n_obj, (sq, skv, skv), (da, da, 96), (0, 1, 0))
    plain = ops.pool_ln_fwd_qkv(*args, sels=sels, out_scales=(1.0, KSC, 1.0))
    qa_ref = plain[0][0].clone()
    qa_ref[..., 96:] = 7.0
    ops.gemm_nt(qa_ref.view(-1, da)[:, :96], rcat, None, hip.EPI_RELQ, relq=(cmap, qa_ref, 1.4426950408889634))
    fused = ops.pool_ln_fwd_qkv(*args, sels=sels, out_scales=(1.0, KSC, 1.0), relq=(rcat, cmap, 1.4426950408889634))
    assert torch.equal(fused[0][0], qa_ref)
    for i in (1, 2):
        assert torch.equal(fused[i][0], plain[i][0])
    for i in range(3):
        for a_, b_ in zip(fused[i][1:], plain[i][1:]):
            assert torch.equal(a_, b_)


@pytest.mark.parametrize("B,L,Tx,O,C", [(3, 37, 8, 4, 96), (2, 5, 1, 3, 192)])
def test_special_token_grads(ops, B, L, Tx, O, C):
    """svit_special_token_grads: gradients of cls_token / object_queries / pos_embed_temporal from d(block-0 input)
    in one launch (video_model_builder.py:326-363 backward), ACCUMULATED into the given buffers."""
    N = 1 + L + Tx * O
    dx = rnd("sg%d%d" % (B, Tx), (B, N, C), 1.0)
    g_cls, g_obj = torch.full((1, 1, C), 0.5, device=DEV), torch.full((1, O, C), 0.5, device=DEV)
    g_pos = torch.full((1, Tx, C), 0.5, device=DEV) if Tx > 1 else None
    ops.special_token_grads(dx, g_cls, g_obj, g_pos, L, Tx, O, Tx > 1)
    dobj = dx[:, 1 + L:].reshape(B, Tx, O, C)
    assert rel_err(g_cls - 0.5, dx[:, 0].sum(0).view(1, 1, C)) < 1e-6
    assert rel_err(g_obj - 0.5, dobj.sum((0, 1)).view(1, O, C)) < 1e-6
    if Tx > 1:
        assert rel_err(g_pos - 0.5, dobj.sum((0, 2)).view(1, Tx, C)) < 1e-6


@pytest.mark.parametrize("T,O,drop,which,B", [(16, 4, True, "all", 3), (2, 3, False, "logits", 3), (1, 4, True, "image", 63),
                                               (1, 4, True, "all", 21)])
def test_head_fused_vs_aten(ops, T, O, drop, which, B):
    """svit_head_fwd / svit_head_bwd (one launch each way) against the ATen head they replace in training mode
    (slowfast/models/video_model_builder.py:505-551): all four outputs, d(tokens) incl. the zero rows of the
    patch tokens, and the eight parameter gradients, for the video loss (logits only), the image losses (boxes
    + contact only) and everything at once; with and without dropout factors."""
    C_, n_cls = 768, 174
    N = 1 + 37 + T * O
    tokens = rnd("ht%d" % T, (B, N, C_), 1.0).requires_grad_(True)
    R_ = 1 + T * O
    keep = None
    if drop:
        keep = (torch.rand((B, R_, C_), device=DEV, generator=torch.Generator(DEV).manual_seed(5)) > 0.5).float() * 2.0
    ws = [rnd("hw%d%d" % (i, T), (n, C_), 0.05).requires_grad_(True) for i, n in enumerate((n_cls, 4, 1, 5))]
    bs = [rnd("hb%d%d" % (i, T), (n,), 0.1).requires_grad_(True) for i, n in enumerate((n_cls, 4, 1, 5))]
    # reference: the ATen ops of SViTHead.forward
    feat = torch.cat((tokens[:, :1], tokens[:, -T * O:]), dim=1)
    x = feat * keep if keep is not None else feat
    cls, xo = x[:, 0], x[:, 1:].reshape(B, T, O, C_)
    r_logits = F.linear(cls, ws[0], bs[0])
    r_boxes = torch.cat((F.linear(xo, ws[2], bs[2]), torch.sigmoid(F.linear(xo, ws[1], bs[1]))), dim=-1)
    r_contact = F.linear(xo[:, :, :2], ws[3], bs[3])
    params = [(w.detach(), b.detach()) for w, b in zip(ws, bs)]
    logits, boxes, contact, xobj = ops.head_fwd(tokens.detach(), T, O, keep, params)
    assert rel_err(logits, r_logits) < 1e-5 and rel_err(boxes, r_boxes) < 1e-5 and rel_err(contact, r_contact) < 1e-5
    assert torch.equal(xobj, xo.detach())
    gl, gb, gc, gx = (rnd("hg%d%s" % (i, which), tuple(t.shape), 1.0) for i, t in enumerate((r_logits, r_boxes, r_contact, xo)))
    if which == "logits":
        gb = gc = gx = None
    elif which == "image":
        gl = gx = None
    loss = sum((r * g).sum() for r, g in zip((r_logits, r_boxes, r_contact, xo), (gl, gb, gc, gx)) if g is not None)
    loss.backward()
    pg = [(torch.full_like(w, 0.25), torch.full_like(b, 0.25)) for w, b in zip(ws, bs)]      # += semantics
    dtok = ops.head_bwd(tokens.detach(), T, O, keep, params, boxes, (gl, gb, gc, gx), pg)
    assert rel_err(dtok, tokens.grad) < 1e-5
    assert float(dtok[:, 1:N - T * O].abs().max()) == 0.0
    for (gw, gbias), w, b in zip(pg, ws, bs):
        wg = w.grad if w.grad is not None else torch.zeros_like(w)
        bg = b.grad if b.grad is not None else torch.zeros_like(b)
        assert rel_err(gw - 0.25, wg) < 1e-4 or float(wg.abs().max()) == 0.0 and float((gw - 0.25).abs().max()) == 0.0
        assert rel_err(gbias - 0.25, bg) < 1e-4 or float(bg.abs().max()) == 0.0 and float((gbias - 0.25).abs().max()) == 0.0


@pytest.mark.parametrize("q_thw,k_thw,h", [((8, 14, 14), (8, 7, 7), 4), ((2, 28, 28), (2, 7, 7), 2), ((1, 5, 3), (1, 5, 3), 1)])
def test_attention_bwd_writes_the_relpos_scatter_matrix(ops, q_thw, k_thw, h):
    """svit_attn_bwd_args.relD: the dq kernel's epilogue builds the rel-pos backward's scattered matrix D
    (zero rows for cls / objects, d(relq) * scale at the mapped columns) -- BIT-identical to svit_relpos_scatter
    run on the dqa the same launch wrote, for one- and multi-pass row widths and a ragged last query tile."""
    from svit_amd.engine import rel_sections
    B, O = 2, 3
    Lq, Lk = q_thw[0] * q_thw[1] * q_thw[2], k_thw[0] * k_thw[1] * k_thw[2]
    Nq, Nk, J = 1 + Lq + O, 1 + Lk + O, sum(k_thw)
    DA = 128 if J <= 32 else 160
    qa = rnd("sq%d" % Lq, (B, h, Nq, DA), 1.0, BF16)
    ka = (rnd("sk%d" % Lq, (B, h, Nk, DA), 1.0) * 0.15).to(BF16)
    v = rnd("sv%d" % Lq, (B, h, Nk, 96), 1.0, BF16)
    ctx, lse2 = ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=J)
    dctx = rnd("sd%d" % Lq, tuple(ctx.shape), 1.0, BF16)
    rows = [2 * max(q_thw[i], k_thw[i]) - 1 for i in (1, 2, 0)]
    offs, lpad = rel_sections(rows)
    idx = [R.rel_index(q_thw[1], k_thw[1]), R.rel_index(q_thw[2], k_thw[2]), R.rel_index(q_thw[0], k_thw[0])]
    idx_d = [t.to(torch.int32).to(DEV).contiguous() for t in idx]
    kt, kh, kw = k_thw
    body = torch.full((q_thw[0], q_thw[1], q_thw[2], DA - 96), -1, dtype=torch.int32)
    body[..., :kh] = (offs[0] + idx[0].to(torch.int32)).view(1, q_thw[1], 1, kh)
    body[..., kh:kh + kw] = (offs[1] + idx[1].to(torch.int32)).view(1, 1, q_thw[2], kw)
    body[..., kh + kw:J] = (offs[2] + idx[2].to(torch.int32)).view(q_thw[0], 1, 1, kt)
    cmap = torch.full((Nq, DA - 96), -1, dtype=torch.int32)
    cmap[1:1 + Lq] = body.view(Lq, DA - 96)
    from svit_amd import hip
    rt = rnd("srt%d" % Lq, (96, lpad), 0.3, BF16)
    dqa, dk, dv, D, X = ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, 96 ** -0.5, bias_cols=J,
                                     reld=(cmap.to(DEV).contiguous(), lpad, 1.4426950408889634, rt))
    D_ref = ops.relpos_scatter(dqa, idx_d, offs, lpad, B, h, q_thw, k_thw, O, 1.4426950408889634)
    assert torch.equal(D, D_ref)
    # ... and (narrow tables only) dq_extra = D . rt^T from the same launch, against the GEMM it replaces
    assert (X is None) == (lpad > 128)
    if X is not None:
        X_ref = ops.gemm_nt(D_ref, rt, None, hip.EPI_F32)
        assert rel_err(X, X_ref) < 1e-5
        # fold mode: the same product added into dqa's first 96 columns inside the kernel (one rounding to bf16
        # instead of two); D and the extra columns stay bit-identical
        dqa_f, dk_f, dv_f, D_f, tag = ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, 96 ** -0.5, bias_cols=J,
                                                   reld=(cmap.to(DEV).contiguous(), lpad, 1.4426950408889634, rt, "fold"))
        assert tag == "folded" and torch.equal(D_f, D_ref) and torch.equal(dqa_f[..., 96:], dqa[..., 96:])
        want = dqa[..., :96].float() + X_ref.view(B, h, Nq, 96)
        assert rel_err(dqa_f[..., :96], want) < 1e-2 and cos(dqa_f[..., :96], want) > 0.99999
        assert torch.equal(dk_f, dk) and torch.equal(dv_f, dv)
    dqa2, dk2, dv2 = ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, 96 ** -0.5, bias_cols=J)
    assert torch.equal(dqa, dqa2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)


@pytest.mark.parametrize("q_thw,k_thw", [((2, 8, 8), (2, 2, 2)), ((2, 4, 4), (2, 4, 4)),
                                         ((3, 5, 5), (3, 3, 3)), ((1, 4, 4), (1, 2, 2)),
                                         ((2, 14, 14), (2, 14, 14)), ((2, 56, 56), (2, 7, 7))])
def test_relpos_q(ops, q_thw, k_thw):
    B, h, O = 2, 2, 3
    Lq = q_thw[0] * q_thw[1] * q_thw[2]
    J = sum(k_thw)
    ld = 128 if J <= 32 else 160
    qa = torch.zeros((B, h, 1 + Lq + O, ld), device=DEV, dtype=BF16)
    qa[..., :96] = rnd("rq%d" % Lq, (B, h, 1 + Lq + O, 96), 1.0, BF16)
    qa[..., 96:] = 5.0  # must be overwritten
    rows = [2 * max(q_thw[i], k_thw[i]) - 1 for i in (1, 2, 0)]
    tabs = [rnd("rt%d%d" % (i, Lq), (rows[i], 96), 0.3) for i in range(3)]
    idx = [R.rel_index(q_thw[1], k_thw[1]), R.rel_index(q_thw[2], k_thw[2]),
           R.rel_index(q_thw[0], k_thw[0])]
    idx_d = [t.to(torch.int32).to(DEV).contiguous() for t in idx]
    scale = 96 ** -0.5
    ops.relpos_q_fwd(qa, tabs, idx_d, B, h, q_thw, k_thw, O, 1.0 / scale)
    q = qa[..., :96].float().cpu().requires_grad_(True)
    tc = [t.cpu().requires_grad_(True) for t in tabs]
    bias = R.rel_pos_bias(q, q_thw, k_thw, tc[0], tc[1], tc[2])        # [B,h,Lq,Lk]
    # rebuild the bias from the stored per-query vectors and the key coordinates
    rel = qa[..., 96:].float().cpu() * scale
    kt, kh, kw = k_thw
    p = torch.arange(kt * kh * kw)
    got = (rel[:, :, 1:1 + Lq][..., (p // kw) % kh] + rel[:, :, 1:1 + Lq][..., kh + p % kw] +
           rel[:, :, 1:1 + Lq][..., kh + kw + p // (kw * kh)])
    assert rel_err(got, bias) < 2e-2 and cos(got, bias) > 0.9999
    assert float(rel[:, :, 0].abs().max()) == 0 and float(rel[:, :, 1 + Lq:].abs().max()) == 0
    assert float(rel[..., J:].abs().max()) == 0
    # backward
    dqa = torch.zeros_like(qa)
    dqa[..., 96:96 + J] = rnd("rdq%d" % Lq, (B, h, 1 + Lq + O, J), 1.0, BF16)
    dtabs = [torch.zeros_like(t) for t in tabs]
    dq_extra = ops.relpos_q_bwd(qa, dqa, tabs, idx_d, dtabs, B, h, q_thw, k_thw, O, 1.0 / scale)
    # reference: stored value s_j = relq_j/scale  =>  loss = sum_j d_j * relq_j / scale
    d = dqa[..., 96:96 + J].float().cpu()[:, :, 1:1 + Lq] / scale
    qp = q[:, :, 1:1 + Lq].reshape(B, h, q_thw[0], q_thw[1], q_thw[2], 96)
    Rh, Rw, Rt = tc[0][idx[0]], tc[1][idx[1]], tc[2][idx[2]]
    relq = torch.cat([torch.einsum("bntyxc,ykc->bntyxk", qp, Rh),
                      torch.einsum("bntyxc,xkc->bntyxk", qp, Rw),
                      torch.einsum("bntyxc,tkc->bntyxk", qp, Rt)], dim=-1).reshape(B, h, Lq, J)
    (relq * d).sum().backward()
    assert rel_err(dq_extra, q.grad) < 2e-3
    for i in range(3):
        assert rel_err(dtabs[i], tc[i].grad) < 2e-3
    # forward, GEMM formulation the engine uses: P = q Rcat^T, then gather
    from svit_amd import hip
    rows_off = (0, rows[0], rows[0] + rows[1])
    lp96 = (sum(rows) + 95) // 96 * 96
    rc16 = torch.zeros((lp96, 96), device=DEV, dtype=BF16)
    rc16[:sum(rows)] = torch.cat(tabs, 0).to(BF16)
    qb = qa.clone()
    qb[..., 96:] = 5.0
    Pm = ops.gemm_nt(qb.view(-1, ld)[:, :96], rc16, None, hip.EPI_BF16)
    ops.relpos_gather(Pm, qb, idx_d, rows_off, B, h, q_thw, k_thw, O, 1.0 / scale)
    rel2 = qb[..., 96:].float().cpu() * scale
    got2 = (rel2[:, :, 1:1 + Lq][..., (p // kw) % kh] + rel2[:, :, 1:1 + Lq][..., kh + p % kw] +
            rel2[:, :, 1:1 + Lq][..., kh + kw + p // (kw * kh)])
    assert rel_err(got2, bias) < 3e-2 and cos(got2, bias) > 0.9998
    assert float(rel2[:, :, 0].abs().max()) == 0 and float(rel2[..., J:].abs().max()) == 0
    # round 3: the same in ONE launch (EPI_RELQ: the GEMM's epilogue does the gather, P is never stored) --
    # bit-identical to the pair above; the map is what Engine._relq_map builds
    Nq_ = 1 + Lq + O
    cmap = torch.full((Nq_, ld - 96), -1, dtype=torch.int32)
    body = torch.full((q_thw[0], q_thw[1], q_thw[2], ld - 96), -1, dtype=torch.int32)
    body[..., :kh] = (rows_off[0] + idx[0].to(torch.int32)).view(1, q_thw[1], 1, kh)
    body[..., kh:kh + kw] = (rows_off[1] + idx[1].to(torch.int32)).view(1, 1, q_thw[2], kw)
    body[..., kh + kw:J] = (rows_off[2] + idx[2].to(torch.int32)).view(q_thw[0], 1, 1, kt)
    cmap[1:1 + Lq] = body.view(Lq, ld - 96)
    qc = qa.clone()
    qc[..., 96:] = 5.0
    ops.gemm_nt(qc.view(-1, ld)[:, :96], rc16, None, hip.EPI_RELQ, relq=(cmap.to(DEV).contiguous(), qc, 1.0 / scale))
    assert torch.equal(qc, qb)
    # the GEMM formulation the engine uses: scatter -> D, dR = D^T q, dq = D Rcat
    from svit_amd.engine import rel_sections
    offs, lpad = rel_sections(rows)
    D = ops.relpos_scatter(dqa, idx_d, offs, lpad, B, h, q_thw, k_thw, O, 1.0 / scale)
    qa2 = qa.view(-1, ld)
    rcat = torch.zeros((lpad, 96), device=DEV)
    for o, tb in zip(offs, tabs):
        rcat[o:o + tb.shape[0]] = tb
    for i in range(3):
        d = torch.zeros_like(tabs[i])
        ops.gemm_tn(D[:, offs[i]:offs[i] + rows[i]], qa2[:, :96], d)
        assert rel_err(d, tc[i].grad) < 1e-2 and cos(d, tc[i].grad) > 0.9999
    dq2 = ops.gemm_nt(D, rcat.t().contiguous().to(BF16), None, hip.EPI_F32)
    assert rel_err(dq2.view(B, h, -1, 96), q.grad) < 1e-2 and cos(dq2.view(B, h, -1, 96), q.grad) > 0.9999



@pytest.mark.parametrize("B,h,thw,sq,skv,n_obj", [
    (2, 4, (8, 14, 14), 1, 2, 64),     # blocks 4-13 of 16x224^2
    (1, 4, (16, 14, 14), 1, 2, 128),   # 32x224^2: three t-chunks per tensor
    (1, 8, (16, 14, 14), 2, 1, 128),   # block 14 at 32 frames
    (3, 4, (1, 14, 14), 1, 2, 4),      # frames pass (T' = 1)
    (2, 8, (8, 7, 7), 1, 1, 64),       # block 15
    (2, 2, (5, 9, 13), 2, 1, 8),       # odd plane, odd T
])
def test_pool_slab_forward_vs_conv3d(ops, B, h, thw, sq, skv, n_obj):
    """Round-3 slab stencil (csrc/pool.hip::pool_slab_fwd_kernel + pool_slab_ln_kernel; the path the
    engine takes on planes <= 14x14): pre-LN values against torch's depthwise conv3d on the same bf16
    operands, object / cls rows against the closed form, and out / mean / rstd against the streaming
    kernels of the same library (svit_debug_set_pool switches the path)."""
    import ctypes as C
    from svit_amd import hip
    lib = hip.load()
    lib.svit_debug_set_pool.restype, lib.svit_debug_set_pool.argtypes = C.c_int32, [C.c_int32, C.c_int32]
    T, H, W = thw
    L = T * H * W
    N = 1 + L + n_obj
    qkv = rnd("sl%d%d%d" % (T, H, h), (B, N, 3, h, 96), 0.5, BF16)
    ws = [rnd("sw%d%d" % (i, H), (96, 27), 0.2) for i in range(3)]
    g = [rnd("sg%d" % i, (96,), 0.3) + 1.0 for i in range(3)]
    b = [rnd("sb%d" % i, (96,), 0.1) for i in range(3)]
    wflat = torch.cat([w.flatten() for w in ws]).contiguous()
    offs = torch.tensor([0, 2592, 5184], dtype=torch.int64, device=DEV)
    sel = ops.pool_weight_sel(wflat, offs, torch.zeros((3, 2592), dtype=torch.int32, device=DEV))
    sels = [sel[i] for i in range(3)]
    J = 2 * ops.pooled(H, skv) + T
    da = 128 if J <= 32 else 160
    res = []
    try:
        for on in (0, 1, 3):       # streaming kernels / VALU slab conv / MFMA conv wherever its geometry holds (else the slab)
            lib.svit_debug_set_pool(0, on)
            r = ops.pool_ln_fwd_qkv(qkv, ws, g, b, B, h, thw, n_obj, (sq, skv, skv), (da, da, 96), (0, 1, 0),
                                    sels=sels, out_scales=(1.0, KSC, 1.0))
            torch.cuda.synchronize()
            res.append(r)
    finally:
        lib.svit_debug_reset()
    for which, s in ((0, sq), (1, skv), (2, skv)):
        x = qkv[:, 1:1 + L, which].float()
        vol = x.reshape(B, T, H, W, h, 96).permute(0, 4, 5, 1, 2, 3).reshape(B * h, 96, T, H, W)
        w = ws[which].to(BF16).float().reshape(96, 1, 3, 3, 3)
        ref = F.conv3d(vol, w, None, stride=(1, s, s), padding=1, groups=96)
        Lo = ref.shape[2] * ref.shape[3] * ref.shape[4]
        ref = ref.reshape(B, h, 96, -1).transpose(2, 3)
        gain = R.object_gain(ws[which].to(BF16).float().cpu().reshape(96, 1, 3, 3, 3), (1, s, s))
        obj = qkv[:, 1 + L:, which].float().cpu().permute(0, 2, 1, 3) * gain
        so, sp, sm, sr = res[0][which]
        cols = slice(0, 96) if which == 0 else slice(None)     # q's bias columns are the gather's to write
        for path in (1, 2):
            out, pre, mean, rstd = res[path][which]
            assert rel_err(pre[:, :, 1:1 + Lo], ref) < 1e-2          # bf16 rounding of the stored value
            assert rel_err(pre[:, :, 1 + Lo:], obj) < 2e-2
            assert torch.equal(pre[:, :, 0].cpu(), qkv[:, 0, which].cpu())
            assert rel_err(out[..., cols], so[..., cols]) < 2e-2
            assert rel_err(mean, sm) < 1e-3 and rel_err(rstd, sr) < 1e-3
        # the MFMA conv (round 4: banded-Toeplitz v_mfma_f32_4x4x4_16b_bf16 products) against the VALU slab conv:
        # the same bf16 operands and fp32 accumulation, another summation order -- the stored bf16 values may differ
        # in the last place on a few elements, nothing more
        pm, ps = res[2][which][1].float(), res[1][which][1].float()
        assert rel_err(pm, ps) < 8e-3
        assert float(((pm - ps).abs() > 0.02 * ps.abs().max()).float().mean()) == 0.0


@pytest.mark.parametrize("B,h,thw,sq,skv,n_obj", [
    (1, 1, (2, 56, 56), 1, 8, 8),       # block 0 (two of its planes): 56x56 cut in y, stride-8 windows packed
    (1, 2, (3, 56, 56), 2, 4, 12),      # block 1
    (2, 2, (3, 28, 28), 1, 4, 12),      # block 2
    (1, 4, (4, 28, 28), 2, 2, 16),      # block 3
    (2, 1, (1, 56, 56), 1, 8, 4),       # image rank, block 0 (T' = 1)
    (1, 2, (2, 19, 26), 1, 3, 8),       # odd plane, stride 3
    (1, 2, (2, 23, 17), 2, 2, 8),       # odd plane, stride 2
])
def test_pool_forward_staged_large_planes(ops, B, h, thw, sq, skv, n_obj):
    """Round 5: the staged conv forward of the planes past 14x14 (csrc/pool.hip::pool_fwd_staged_kernel + the row-wise
    LayerNorm launch) against torch's depthwise conv3d on the same bf16 operands, the closed form of the object / cls rows,
    and the streaming kernel of the same library (svit_debug_set_pool(2, 0)) incl. the rel-pos columns of q."""
    import ctypes as C
    from svit_amd import hip
    lib = hip.load()
    T, H, W = thw
    L = T * H * W
    N = 1 + L + n_obj
    qkv = rnd("fs%d%d%d%d" % (T, H, W, h), (B, N, 3, h, 96), 0.5, BF16)
    ws = [rnd("fw%d%d" % (i, H), (96, 27), 0.2) for i in range(3)]
    g = [rnd("fg%d" % i, (96,), 0.3) + 1.0 for i in range(3)]
    b = [rnd("fb%d" % i, (96,), 0.1) for i in range(3)]
    J = ops.pooled(H, skv) + ops.pooled(W, skv) + T
    da = 128 if J <= 32 else 160
    if J > 64:
        pytest.skip("key grid too large for the in-MFMA rel-pos columns")
    res = []
    try:
        for on in (0, 1):
            assert lib.svit_debug_set_pool(2, on) == 0
            r = ops.pool_ln_fwd_qkv(qkv, ws, g, b, B, h, thw, n_obj, (sq, skv, skv), (da, da, 96), (0, 1, 0),
                                    out_scales=(1.0, KSC, 1.0))
            torch.cuda.synchronize()
            res.append(r)
    finally:
        lib.svit_debug_reset()
    for which, s in ((0, sq), (1, skv), (2, skv)):
        x = qkv[:, 1:1 + L, which].float()
        vol = x.reshape(B, T, H, W, h, 96).permute(0, 4, 5, 1, 2, 3).reshape(B * h, 96, T, H, W)
        w = ws[which].to(BF16).float().reshape(96, 1, 3, 3, 3)
        ref = F.conv3d(vol, w, None, stride=(1, s, s), padding=1, groups=96)
        Lo = ref.shape[2] * ref.shape[3] * ref.shape[4]
        ref = ref.reshape(B, h, 96, -1).transpose(2, 3)
        gain = R.object_gain(ws[which].to(BF16).float().cpu().reshape(96, 1, 3, 3, 3), (1, s, s))
        obj = qkv[:, 1 + L:, which].float().cpu().permute(0, 2, 1, 3) * gain
        so, sp, sm, sr = res[0][which]
        out, pre, mean, rstd = res[1][which]
        assert rel_err(pre[:, :, 1:1 + Lo], ref) < 1e-2          # bf16 rounding of the stored value
        assert rel_err(pre[:, :, 1 + Lo:], obj) < 2e-2
        assert torch.equal(pre[:, :, 0].cpu(), qkv[:, 0, which].cpu())
        assert rel_err(out[..., :96], so[..., :96]) < 2e-2 and cos(out[..., :96], so[..., :96]) > 0.9999
        if which == 1:      # the one-hot key coordinates are copied: exactly equal
            assert torch.equal(out[..., 96:], so[..., 96:])
        assert rel_err(mean, sm) < 1e-3 and rel_err(rstd, sr) < 1e-3
        assert rel_err(pre.float(), sp.float()) < 8e-3


def test_table_interp_matches_matmul(ops):
    """svit_table_interp (rel-pos tables at another resolution: attention.py:84-137's F.interpolate as one matrix) against
    torch's fp32 matmul, fp32 and bf16 outputs, zero pad rows included."""
    m = torch.zeros((96, 27 + 27 + 15), device=DEV)
    m[:55, :27] = rnd("ti0", (55, 27), 1.0).abs()
    m[55:80, 27:54] = rnd("ti1", (25, 27), 1.0)
    m[80:81, 54:] = rnd("ti2", (1, 15), 1.0)
    t = rnd("ti3", (69, 96), 0.7)
    ref = m.double() @ t.double()
    r32, r16 = ops.table_interp(m, t)
    assert rel_err(r32, ref) < 1e-6
    assert torch.equal(r16, r32.to(BF16))
    assert float(r32[81:].abs().max()) == 0.0
    none32, r16b = ops.table_interp(m, t, want_f32=False)
    assert none32 is None and torch.equal(r16b, r16)
    # the batched form (every block of a forward pass in one launch): two jobs of different sizes, one without fp32 output
    m2 = rnd("ti4", (192, 40), 1.0)
    t2 = rnd("ti5", (40, 96), 0.7)
    o32 = torch.full((96, 96), 7.0, device=DEV)
    o16 = torch.empty((96, 96), device=DEV, dtype=BF16)
    p16 = torch.empty((192, 96), device=DEV, dtype=BF16)
    jobs = ops.table_interp_jobs([(m, t, o32, o16), (m2, t2, None, p16)], DEV)
    ops.table_interp_batched(jobs, 192)
    assert torch.equal(o32, r32) and torch.equal(o16, r16)
    assert torch.equal(p16, ops.table_interp(m2, t2)[1])


@pytest.mark.parametrize("B,h,hw,sq,skv,n_obj,save", [
    (6, 1, (56, 56), 1, 8, 4, False),     # frames pass, block 0: bands of 4 output rows, packed stride-8 windows
    (3, 2, (56, 56), 2, 4, 4, False),     # block 1
    (5, 2, (28, 28), 1, 4, 4, True),      # block 2 (an image rank keeps pre / mean / rstd)
    (3, 4, (28, 28), 2, 2, 4, False),     # block 3
    (9, 4, (14, 14), 1, 2, 4, False),     # blocks 4-13: whole planes
    (4, 8, (14, 14), 2, 1, 4, True),      # block 14
    (7, 8, (7, 7), 1, 1, 4, False),       # block 15
    (2, 2, (19, 26), 1, 3, 3, True),      # odd plane, stride 3
    (2, 3, (23, 17), 2, 2, 0, False),     # odd plane, no object tokens
])
def test_pool_forward_one_plane_volumes(ops, B, h, hw, sq, skv, n_obj, save):
    """Round 5: T = 1 volumes (frames pass, image ranks) take csrc/pool.hip::pool_frame_fwd_kernel -- conv + LayerNorm in one
    launch from an LDS-staged plane.  Against torch's depthwise conv3d + layer_norm on the same bf16 operands, the closed
    form of the object / cls rows, and the other paths of the same library (svit_debug_set_pool(3, 0)) incl. q's rel-pos
    columns and the keys' one-hot coordinates; with and without the saved-for-backward trio."""
    from svit_amd import hip
    lib = hip.load()
    H, W = hw
    thw = (1, H, W)
    L = H * W
    N = 1 + L + n_obj
    qkv = rnd("fr%d%d%d%d" % (B, H, W, h), (B, N, 3, h, 96), 0.5, BF16)
    ws = [rnd("frw%d%d" % (i, H), (96, 27), 0.2) for i in range(3)]
    g = [rnd("frg%d" % i, (96,), 0.3) + 1.0 for i in range(3)]
    b = [rnd("frb%d" % i, (96,), 0.1) for i in range(3)]
    J = ops.pooled(H, skv) + ops.pooled(W, skv) + 1
    if J > 64:
        pytest.skip("key grid too large for the in-MFMA rel-pos columns")
    da = 128 if J <= 32 else 160
    res = []
    try:
        for on in (0, 2):        # 2: also where the trio is saved (the default takes the kernel in no-grad passes only)
            assert lib.svit_debug_set_pool(3, on) == 0
            r = ops.pool_ln_fwd_qkv(qkv, ws, g, b, B, h, thw, n_obj, (sq, skv, skv), (da, da, 96), (0, 1, 0),
                                    out_scales=(1.0, KSC, 1.0), save=save)
            torch.cuda.synchronize()
            res.append(r)
    finally:
        lib.svit_debug_reset()
    for which, s in ((0, sq), (1, skv), (2, skv)):
        x = qkv[:, 1:1 + L, which].float()
        vol = x.reshape(B, 1, H, W, h, 96).permute(0, 4, 5, 1, 2, 3).reshape(B * h, 96, 1, H, W)
        w = ws[which].to(BF16).float().reshape(96, 1, 3, 3, 3)
        ref = F.conv3d(vol, w, None, stride=(1, s, s), padding=1, groups=96)
        Lo = ref.shape[3] * ref.shape[4]
        ref = ref.reshape(B, h, 96, -1).transpose(2, 3)
        gain = R.object_gain(ws[which].to(BF16).float().cpu().reshape(96, 1, 3, 3, 3), (1, s, s)).to(ref.device)
        obj = qkv[:, 1 + L:, which].float().permute(0, 2, 1, 3) * gain
        cls = qkv[:, :1, which].float().permute(0, 2, 1, 3)
        pre_ref = torch.cat([cls, ref, obj], dim=2).to(BF16).float()
        sc = KSC if which == 1 else 1.0
        ln_ref = F.layer_norm(pre_ref, (96,), g[which], b[which], 1e-6) * sc
        so, sp, sm, sr = res[0][which]
        out, pre, mean, rstd = res[1][which]
        assert out.shape == so.shape == (B, h, 1 + Lo + n_obj, da if which < 2 else 96)
        assert rel_err(out[..., :96], ln_ref) < 2e-2 and cos(out[..., :96], ln_ref) > 0.9999
        assert rel_err(out[..., :96], so[..., :96]) < 2e-2 and cos(out[..., :96], so[..., :96]) > 0.9999
        if which == 1:      # the one-hot key coordinates: exactly equal
            assert torch.equal(out[..., 96:], so[..., 96:])
        if save:
            assert rel_err(pre[:, :, 1:1 + Lo], ref) < 1e-2
            assert torch.equal(pre[:, :, 0].cpu(), qkv[:, 0, which].cpu())
            assert rel_err(mean, sm) < 1e-3 and rel_err(rstd, sr) < 1e-3
            assert rel_err(pre.float(), sp.float()) < 8e-3
        else:
            assert pre is None and mean is None and rstd is None


# -------------------------------------------------------------------- fused attention ----
KSC = (96 ** -0.5) * math.log2(math.e)   # what the pooling kernel multiplies the keys by (engine.K_SCALE)


def _attn_ref(qa, ka, v, scale):
    """operand convention of svit_attn_fwd (include/svit_hip.h): qa . ka^T is the score in the
    log2 domain (the keys arrive multiplied by scale * log2 e)."""
    s = (qa.float() @ ka.float().transpose(-1, -2)) * math.log(2.0)
    p = s.softmax(-1)
    o = p @ v.float()
    o = torch.cat([o[:, :, :1], o[:, :, 1:] + qa[:, :, 1:, :96].float()], dim=2)
    B, h, Nq, _ = o.shape
    return o.transpose(1, 2).reshape(B, Nq, h * 96), s


@pytest.mark.parametrize("Nq,Nk,DA,h", [(200, 70, 128, 2), (457, 457, 128, 1), (130, 300, 160, 2),
                                        (33, 64, 128, 1), (700, 129, 160, 1)])
def test_attention_fwd_bwd(ops, Nq, Nk, DA, h):
    B = 2
    scale = 96 ** -0.5
    qa = rnd("aq%d" % Nq, (B, h, Nq, DA), 1.0, BF16)
    ka = rnd("ak%d" % Nk, (B, h, Nk, DA), KSC, BF16)
    v = rnd("av%d" % Nk, (B, h, Nk, 96), 1.0, BF16)
    ctx, lse2 = ops.attn_fwd(qa, ka, v, scale)
    qr = qa.float().cpu().requires_grad_(True)
    kr = ka.float().cpu().requires_grad_(True)
    vr = v.float().cpu().requires_grad_(True)
    ref, s = _attn_ref(qr, kr, vr, scale)
    assert rel_err(ctx, ref) < 2e-2 and cos(ctx, ref) > 0.9999
    lse_ref = torch.logsumexp(s, dim=-1) * math.log2(math.e)
    assert rel_err(lse2, lse_ref) < 1e-3
    dctx = rnd("ad%d" % Nq, (B, Nq, h * 96), 1.0, BF16)
    # the kernels do not include the residual-pooling path in dq (it is fed to pool_ln_bwd)
    o_only = ref - torch.cat([torch.zeros(B, 1, h * 96),
                              qr[:, :, 1:, :96].transpose(1, 2).reshape(B, Nq - 1, h * 96)], 1)
    o_only.backward(dctx.float().cpu())
    import ctypes as C
    from svit_amd import hip
    lib = hip.load()
    lib.svit_attn_debug_set.restype, lib.svit_attn_debug_set.argtypes = C.c_int32, [C.c_int32, C.c_int32]
    try:
        for halves in (1, 2):      # dkv kernel: 4 waves / 8 waves (two query halves per tile)
            lib.svit_attn_debug_set(0, halves)
            for splits in (0, 1, 3):
                dqa, dk, dv = ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, scale, q_splits=splits)
                dk, dv = dk.sum(0), dv.sum(0)      # one partial plane per chunk of the query range
                assert cos(dqa, qr.grad) > 0.999 and rel_err(dqa, qr.grad) < 4e-2
                dk_ref = kr.grad[..., :96] * KSC      # dk is taken with respect to the UN-scaled pooled keys
                assert cos(dk, dk_ref) > 0.999 and rel_err(dk, dk_ref) < 4e-2, (halves, splits)
                assert cos(dv, vr.grad) > 0.999 and rel_err(dv, vr.grad) < 4e-2, (halves, splits)
    finally:
        lib.svit_attn_debug_set(0, 0)


@pytest.mark.parametrize("Nk", [64, 128, 130, 457])
def test_attention_fwd_eight_wave_path(ops, Nk):
    """grids of >= 200 256-query workgroups at DA = 160 take the 8-wave / 3-stage forward kernel
    (what the 28x28 -> 14x14 blocks of the B = 8 step run): 1, 2, 3 (ragged) and 8 key tiles."""
    B, h, Nq, DA = 8, 2, 3300, 160
    scale = 96 ** -0.5
    qa = rnd("wq", (B, h, Nq, DA), 1.0, BF16)
    ka = rnd("wk%d" % Nk, (B, h, Nk, DA), KSC, BF16)
    v = rnd("wv%d" % Nk, (B, h, Nk, 96), 1.0, BF16)
    ctx, lse2 = ops.attn_fwd(qa, ka, v, scale)
    ref, s = _attn_ref(qa.cpu(), ka.cpu(), v.cpu(), scale)
    assert rel_err(ctx, ref) < 2e-2 and cos(ctx, ref) > 0.9999
    assert rel_err(lse2, torch.logsumexp(s, dim=-1) * math.log2(math.e)) < 1e-3


@pytest.mark.parametrize("B,h,Nq,Nk,DA,J", [
    (8, 4, 1633, 457, 128, 22),    # blocks 4-13: two query blocks per wave, 8 tiles (ragged), 8 k-steps
    (8, 4, 1633, 1633, 160, 36),   # block 3: 9 of 10 k-steps, 26 tiles
    (2, 8, 457, 457, 128, 22),     # small grid: one query block per wave
    (2, 2, 300, 54, 128, 15),      # frames pass: a single ragged tile, 7 k-steps
    (2, 2, 300, 201, 160, 29),     # frames pass: 4 tiles, J <= 32 at DA 160 -> 8 k-steps
    (1, 1, 70, 64, 128, 32),       # exactly one full tile
    (1, 1, 70, 128, 160, 64),      # two tiles, every bias column in use (10 k-steps)
    (1, 1, 257, 192, 128, 0),      # three tiles; bias_cols unknown (0 = all)
    (1, 2, 129, 330, 128, 22),     # six tiles: both halves of the two-tile unrolled loop end the sweep
])
def test_attention_fwd_pipelined_kernel(ops, B, h, Nq, Nk, DA, J):
    """The round-2 forward kernel (csrc/attn_fwd.hip: buffer-descriptor LDS-DMA, row sums on the
    matrix pipe, only the k-steps that carry data) against the fp32 reference, over the tile-count,
    raggedness and bias-column cases of its pipeline (1, 2, 3, 4, 6, 8, 26 tiles; 7-10 k-steps)."""
    scale = 96 ** -0.5
    qa = rnd("pq%d_%d" % (Nq, DA), (B, h, Nq, DA), 1.0, BF16)
    ka = rnd("pk%d_%d" % (Nk, DA), (B, h, Nk, DA), KSC, BF16)
    v = rnd("pv%d" % Nk, (B, h, Nk, 96), 1.0, BF16)
    if J:
        qa[..., 96 + J:] = 0      # columns past J carry no data (the pool kernel writes zeros)
        ka[..., 96 + J:] = 0
    ctx, lse2 = ops.attn_fwd(qa, ka, v, scale, bias_cols=J)
    ref, s = _attn_ref(qa.float().cpu(), ka.float().cpu(), v.float().cpu(), scale)
    assert rel_err(ctx, ref) < 2e-2 and cos(ctx, ref) > 0.9999
    assert rel_err(lse2, torch.logsumexp(s, dim=-1) * math.log2(math.e)) < 1e-3



@pytest.mark.parametrize("B,h,Nq,Nk,J", [
    (3, 2, 201, 54, 15),       # frames pass, 14x14 stage: two 128-query tiles per (batch, head)
    (2, 1, 3141, 54, 15),      # frames pass, 56x56 stage: workgroups walk several tiles
    (5, 8, 54, 54, 15),        # frames pass, last stage: one ragged tile
    (2, 2, 130, 20, 9),        # at most 32 keys: half a tile
    (1, 2, 789, 64, 0),        # exactly one full tile, every bias column carries data
    (2, 3, 257, 33, 30),       # KSU = 8
])
def test_attention_fwd_short_key_tile(ops, B, h, Nq, Nk, J):
    """Round 4, the T' = 1 tile (csrc/attn_fwd.hip::attn_fwd_short_kernel): Nk <= 64 runs a kernel of
    its own -- K / V once per workgroup, waves persistent over 128-query tiles without further barriers.  Its arithmetic is
    the generic kernel's first tile operation for operation: BIT-identical outputs (svit_attn_debug_set(3, 0) selects the
    generic kernel), and the fp32 reference within the usual tolerance."""
    from svit_amd import hip
    lib = hip.load()
    scale = 96 ** -0.5
    DA = 128
    qa = rnd("sq%d_%d" % (Nq, Nk), (B, h, Nq, DA), 1.0, BF16)
    ka = rnd("sk%d_%d" % (Nq, Nk), (B, h, Nk, DA), KSC, BF16)
    v = rnd("sv%d_%d" % (Nq, Nk), (B, h, Nk, 96), 1.0, BF16)
    if J:
        qa[..., 96 + J:] = 0
        ka[..., 96 + J:] = 0
    try:
        assert lib.svit_attn_debug_set(3, 0) == 0
        ctx0, lse0 = ops.attn_fwd(qa, ka, v, scale, bias_cols=J)
        torch.cuda.synchronize()
        assert lib.svit_attn_debug_set(3, 1) == 0
        ctx = torch.full_like(ctx0, float("nan"))
        ctx, lse2 = ops.attn_fwd(qa, ka, v, scale, bias_cols=J)
        torch.cuda.synchronize()
    finally:
        lib.svit_attn_debug_set(3, 1)
    assert torch.equal(ctx, ctx0) and torch.equal(lse2, lse0)
    ref, s = _attn_ref(qa.float().cpu(), ka.float().cpu(), v.float().cpu(), scale)
    assert rel_err(ctx, ref) < 2e-2 and cos(ctx, ref) > 0.9999
    assert rel_err(lse2, torch.logsumexp(s, dim=-1) * math.log2(math.e)) < 1e-3


@pytest.mark.parametrize("Nk,DA", [(9, 128), (54, 128), (100, 160), (457, 128)])
def test_attention_ragged_tile_reads_nothing_past_the_keys(ops, Nk, DA):
    """K and V are views into larger NaN-filled buffers (including behind the LAST (batch, head)):
    the ragged last tile must not pick up a single element from past row Nk, in the forward and in
    both backward kernels (round 2 relied on an LDS-DMA zero-filling out-of-range rows; the scalar
    offset of a buffer descriptor takes no part in its range check)."""
    B, h, Nq = 2, 2, 150
    scale = 96 ** -0.5
    qa = rnd("nq%d" % Nk, (B, h, Nq, DA), 1.0, BF16)
    kbuf = torch.full((B * h * Nk * DA + 64 * DA,), float("nan"), device=DEV, dtype=BF16)
    vbuf = torch.full((B * h * Nk * 96 + 64 * 96,), float("nan"), device=DEV, dtype=BF16)
    ka = kbuf[:B * h * Nk * DA].view(B, h, Nk, DA)
    v = vbuf[:B * h * Nk * 96].view(B, h, Nk, 96)
    ka.copy_(rnd("nk%d" % Nk, (B, h, Nk, DA), KSC, BF16))
    v.copy_(rnd("nv%d" % Nk, (B, h, Nk, 96), 1.0, BF16))
    ctx, lse2 = ops.attn_fwd(qa, ka, v, scale)
    ref, s = _attn_ref(qa.cpu(), ka.cpu(), v.cpu(), scale)
    assert bool(torch.isfinite(ctx.float()).all()) and bool(torch.isfinite(lse2).all())
    assert rel_err(ctx, ref) < 2e-2
    dctx = rnd("nd%d" % Nk, (B, Nq, h * 96), 1.0, BF16)
    dqa, dk, dv = ops.attn_bwd(qa, ka, v, ctx, dctx, lse2, scale)
    for t in (dqa, dk, dv):
        assert bool(torch.isfinite(t.float()).all())


def test_attention_large_scores(ops):
    """Online-softmax rescale path: one key dominates late in the sweep (forces max jumps)."""
    B, h, Nq, Nk, DA = 1, 1, 64, 200, 128
    scale = 96 ** -0.5
    qa = rnd("lq", (B, h, Nq, DA), 1.0, BF16)
    ka = rnd("lk", (B, h, Nk, DA), KSC, BF16)
    ka[:, :, 150] = qa[:, :, 5] * (6 * KSC)
    ka[:, :, 199] = qa[:, :, 9] * (9 * KSC)
    v = rnd("lv", (B, h, Nk, 96), 1.0, BF16)
    ctx, _ = ops.attn_fwd(qa, ka, v, scale)
    ref, _ = _attn_ref(qa.cpu(), ka.cpu(), v.cpu(), scale)
    assert rel_err(ctx, ref) < 2e-2


# ------------------------------------------------------------------- max-pool skip ------
@pytest.mark.parametrize("thw", [(2, 8, 8), (2, 7, 7), (1, 5, 6)])
def test_maxpool(ops, thw):
    B, O, C = 2, 3, 192
    N = 1 + thw[0] * thw[1] * thw[2] + O
    x = rnd("mp%d" % thw[1], (B, N, C), 1.0)
    y, idx = ops.maxpool_fwd(x, thw, O)
    xr = x.cpu().requires_grad_(True)
    ref = R.maxpool_skip(xr, thw, (1, 2, 2), O)
    assert torch.equal(y.cpu(), ref)
    dy = rnd("mpd%d" % thw[1], tuple(ref.shape), 1.0)
    ref.backward(dy.cpu())
    dx = ops.maxpool_bwd(dy, idx, thw, O)
    assert rel_err(dx, xr.grad) < 1e-6
    # bf16 output (the dim-change blocks): the same sums, rounded once -- bit for bit the cast of the f32 result
    dx16 = ops.maxpool_bwd(dy, idx, thw, O, bf16=True)
    assert torch.equal(dx16, ops.scale_cast(dx))


# ------------------------------------------------------------------ optimiser tail ------
def test_clip_adamw(ops):
    n = 100003
    p = rnd("op", (n,), 1.0)
    g = rnd("og", (n,), 0.05)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pw = {"w": p.cpu().clone()}
    st = {"w": (torch.zeros(n), torch.zeros(n))}
    for step in (1, 2):
        ss = torch.zeros(1, device=DEV)
        ops.sumsq(g, ss)
        assert abs(float(ss) - float((g.double() ** 2).sum())) / float(ss) < 1e-5
        ops.adamw_step(p, g, m, v, ss, 1.0, 1e-3, 0.9, 0.999, 1e-8, 1e-4, step)
        R.clip_and_adamw_step(pw, {"w": g.cpu()}, st, 1e-3, step, 1.0, lambda n_, s_: 1e-4)
        assert float((p.cpu() - pw["w"]).abs().max()) < 2e-6


# ------------------------------------------------------------ image-rank HAOG losses ----
@pytest.mark.parametrize("case", ["mixed", "all_empty", "ties"])
def test_haog_loss_fused(ops, case):
    """svit_haog_loss (+_bwd) against the boolean-index torch formulation of the reference
    (svit_amd.losses host path == slowfast/models/losses.py:50-93,138-155), values and grads."""
    from svit_amd import config, losses
    cfg = config.ssv2_cfg(num_frames=4, crop=64)
    B, T = 7, 1
    g = torch.Generator().manual_seed(11)
    boxes = torch.rand(B, T, 4, 4, generator=g) * 0.5 + 0.2
    logit = torch.randn(B, T, 4, 1, generator=g)
    contact = torch.randn(B, T, 2, 5, generator=g)
    tar = torch.rand(B, T, 4, 4, generator=g) * 0.5 + 0.2
    tar[torch.rand(B, T, 4, generator=g) > 0.6] = 0.0
    ctar = torch.randint(-1, 5, (B, 2), generator=g)
    if case == "all_empty":
        tar.zero_()
        ctar.fill_(-1)
    if case == "ties":                      # prediction == target: min/max ties, |d| = 0
        tar[0] = boxes[0]
        tar[1, 0, :, :2] = boxes[1, 0, :, :2]
    res = {}
    for dev in ("cpu", DEV):
        pb = torch.cat((logit, boxes), -1).to(dev).requires_grad_(True)
        pc = contact.clone().to(dev).requires_grad_(True)
        fn = losses.VideoImageLoss(cfg, is_video_rank=False)
        d = fn(None, {"pred_bboxes": pb, "pred_contact_state": pc}, None,
               {"haog_bboxes": tar.to(dev), "contact_state": ctar.to(dev)})
        total = fn.total(d)
        if total.requires_grad:
            total.backward()
        res[dev] = ({k: float(v.detach()) for k, v in d.items()}, float(total.detach()),
                    pb.grad.cpu() if pb.grad is not None else torch.zeros_like(pb).cpu(),
                    pc.grad.cpu() if pc.grad is not None else torch.zeros_like(pc).cpu(), fn)
    ref, got = res["cpu"], res[DEV]
    for k, v in ref[0].items():
        assert abs(got[0][k] - v) <= 2e-6 + 1e-5 * abs(v), (k, got[0][k], v)
    assert abs(got[1] - ref[1]) <= 1e-5 * max(1.0, abs(ref[1]))
    assert float((got[2] - ref[2]).abs().max()) <= 1e-6 + 1e-5 * float(ref[2].abs().max())
    assert float((got[3] - ref[3]).abs().max()) <= 1e-6 + 1e-5 * float(ref[3].abs().max())
    stats = got[4].last_stats.cpu()
    valid = int((~(tar == 0).all(-1)).sum())
    assert int(stats[0]) == valid and int(stats[1]) == int((ctar >= 0).sum()) and int(stats[2]) == 0
    if case == "all_empty":
        assert got[0]["boxes_l1_loss"] == 0.0 and got[0]["boxes_giou_loss"] == 0.0
        assert got[0]["loss_contact_state"] == 0.0 and float(got[3].abs().max()) == 0.0
        assert float(got[2][..., 1:].abs().max()) == 0.0 and float(got[2][..., 0].abs().max()) > 0


# ------------------------------------------------- uint8 input fused into the patch embed ----
def _ref_normalize(u8, mean, std):
    """slowfast/datasets/utils.py:287-303 (tensor_normalize), then T H W C -> C T H W."""
    t = u8.float()
    t = t / 255.0
    t = t - torch.tensor(mean)
    t = t / torch.tensor(std)
    return t.permute(0, 4, 1, 2, 3).contiguous()        # [V,3,T,H,W]


@pytest.mark.parametrize("V,T,Hs,Ws,S,table", [
    (2, 4, 70, 93, 64, [(0, 3, 5), (1, 0, 29), (0, 6, 0), (1, 6, 29)]),
    (1, 2, 312, 415, 312, [(0, 0, 0), (0, 0, 52), (0, 0, 103)]),      # C5: three 312^2 crops, 2 chunks/row
    (3, 1, 64, 64, 64, None),                                           # stills, identity crops
    (1, 3, 41, 59, 37, [(0, 4, 22), (0, 1, 0)]),                        # odd sizes
])
def test_im2col_patch_u8_bit_exact(ops, V, T, Hs, Ws, S, table):
    from svit_amd.input import U8Clips
    g = torch.Generator().manual_seed(V * 1000 + S)
    u8 = torch.randint(0, 256, (V, T, Hs, Ws, 3), generator=g, dtype=torch.uint8)
    mean, std = [0.45, 0.40, 0.5], [0.225, 0.25, 0.2]
    clips = U8Clips(u8.cuda(), S, None if table is None else torch.tensor(table, dtype=torch.int32),
                    mean=mean, std=std)
    cols, thw = ops.im2col_patch_u8(clips)
    f32 = _ref_normalize(u8, mean, std)
    tab = table if table is not None else [(v, 0, 0) for v in range(V)]
    crops = torch.stack([f32[v, :, :, y:y + S, x:x + S] for v, y, x in tab]).cuda().contiguous()
    ref_cols, ref_thw = ops.im2col_patch(crops)
    assert thw == ref_thw and clips.shape == crops.shape
    assert torch.equal(cols.view(torch.int16), ref_cols.view(torch.int16))
    with pytest.raises(ValueError):
        U8Clips(u8.cuda(), S, torch.tensor([[0, Hs - S + 1, 0]], dtype=torch.int32))
    with pytest.raises(ValueError):
        U8Clips(u8.float().cuda(), S)


# ------------------------------------------------------------ round 6: CE loss / the step's random draws ----
@pytest.mark.parametrize("B,C", [(8, 174), (3, 174), (9, 5), (130, 1000)])
def test_ce_loss_fused(ops, B, C):
    """svit_ce_loss (csrc/loss.hip): nn.CrossEntropyLoss(reduction="mean") of VideoImageLoss (losses.py:121,158) and its gradient in one
    launch, against F.cross_entropy on the same fp32 logits -- incl. ignored rows (label -100, torch's default ignore_index) and, through
    svit_amd.losses.cross_entropy, an upstream gradient other than 1."""
    from svit_amd import losses
    x = rnd("ce%d_%d" % (B, C), (B, C), 3.0)
    y = torch.from_numpy(np.arange(B) * 7 % C).to(DEV)
    if B > 4:
        y[2] = -100
    xr = x.cpu().clone().requires_grad_(True)
    ref = F.cross_entropy(xr, y.cpu())
    (ref * 0.37).backward()
    loss, dl = ops.ce_loss(x, y)
    assert abs(float(loss) - float(ref)) < 2e-6 * max(1.0, abs(float(ref)))
    assert rel_err(dl * 0.37, xr.grad) < 1e-5
    xg = x.clone().requires_grad_(True)
    out = losses.cross_entropy(xg, y)
    (out * 0.37).backward()
    assert float(out) == float(loss) and rel_err(xg.grad, xr.grad) < 1e-5
    # a label outside [0, C) that is not the ignore index poisons the loss (fail loudly), every row ignored too
    y2 = y.clone(); y2[0] = C
    assert bool(torch.isnan(ops.ce_loss(x, y2)[0]))
    assert bool(torch.isnan(ops.ce_loss(x, torch.full_like(y, -100))[0]))


def test_step_draws(ops):
    """svit_step_draws (csrc/loss.hip): stochastic-depth factors floor(keep + U) / keep (common.py:46-59) and the head's dropout factors in
    one launch -- values, means, the draw number advancing by itself (also under HIP-graph replay), reproducibility from (seed, draw)."""
    keep = torch.tensor([1.0, 0.9, 0.6, 0.75], device=DEV)
    st = torch.tensor([1234, 0, 0], dtype=torch.int64, device=DEV)
    per, nd, p = 4096, 65 * 768 * 3, 0.5
    s0, d0 = ops.step_draws(st, keep, per, nd, p)
    assert st.tolist() == [1234, 1, 0]
    assert bool((s0[0] == 1.0).all())
    for b in range(1, 4):
        k = float(keep[b])
        on = s0[b] != 0
        assert bool(((s0[b] == 0) | ((s0[b] - 1.0 / k).abs() < 1e-6)).all())
        assert abs(float(on.float().mean()) - k) < 4 * (k * (1 - k) / per) ** 0.5 + 1e-3
    assert bool(((d0 == 0) | (d0 == 2.0)).all()) and abs(float((d0 != 0).float().mean()) - 0.5) < 5e-3
    s1, d1 = ops.step_draws(st, keep, per, nd, p)
    assert st.tolist() == [1234, 2, 0] and not torch.equal(s1, s0) and not torch.equal(d1, d0)
    # same (seed, draw number) -> same numbers; the drop stream does not depend on how many scale values precede it being a multiple of 4
    st2 = torch.tensor([1234, 0, 0], dtype=torch.int64, device=DEV)
    s2, d2 = ops.step_draws(st2, keep, per, nd, p)
    assert torch.equal(s2, s0) and torch.equal(d2, d0)
    st3 = torch.tensor([99, 0, 0], dtype=torch.int64, device=DEV)
    assert not torch.equal(ops.step_draws(st3, keep, per, nd, p)[0], s0)
    # successive draws are uncorrelated (|corr| of two 196k-element masks ~ 1 / sqrt(n))
    a, b = (d0 != 0).float() - 0.5, (d1 != 0).float() - 0.5
    assert abs(float((a * b).mean()) / 0.25) < 0.02
    # under HIP-graph replay every replay draws fresh numbers
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.step_draws(st, keep, per, nd, p)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            sg, dg = ops.step_draws(st, keep, per, nd, p)
    torch.cuda.current_stream().wait_stream(side)
    before = int(st[1])
    g.replay(); torch.cuda.synchronize(); r1 = dg.clone()
    g.replay(); torch.cuda.synchronize(); r2 = dg.clone()
    assert int(st[1]) == before + 2 and not torch.equal(r1, r2) and abs(float((r2 != 0).float().mean()) - 0.5) < 5e-3
