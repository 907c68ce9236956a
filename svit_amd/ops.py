"""Thin tensor-level wrappers over the C ABI (svit_amd/hip.py).

Every function enqueues HIP kernels on torch's current stream and returns torch tensors that
merely *own the memory*; no arithmetic here is done by PyTorch.
"""
import ctypes as C

import torch

from . import hip
from .hip import ptr

BF16 = torch.bfloat16
F32 = torch.float32
HD = 96
_scratch = {}


def scratch(device, floats=8 * 1024 * 1024, tag="main"):
    """Per-device fp32 scratch for the two-stage parameter-gradient reductions (32 MB); `tag`
    names an independent buffer for work that runs on another stream."""
    key = (device.index, floats, tag)
    buf = _scratch.get(key)
    if buf is None:
        buf = torch.empty(floats, device=device, dtype=F32)
        _scratch[key] = buf
    return buf


def _chk_rows(*ts):
    """2-D operands that may be column slices of a wider matrix (unit inner stride)."""
    for t in ts:
        if not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
            raise hip.SvitHipError("svit_amd GEMM operands must be 2-D device tensors with unit inner stride")


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise hip.SvitHipError("svit_amd ops need device tensors (got %s)" % t.device)
        if t is not None and not t.is_contiguous():
            raise hip.SvitHipError("svit_amd ops need contiguous tensors")


def gemm_nt(a, w, bias=None, epilogue=hip.EPI_BF16, out=None, out2=None, aux=None,
            row_scale=None, rows_per_sample=0, accumulate=False, remap=None, save=True, relq=None):
    """C = A[M,K] @ W[N,K]^T with a fused epilogue (see include/svit_hip.h).  `a` may be a
    column slice of a wider matrix (row-strided view).  epilogue EPI_RELQ: relq = (map i32 [Nq, extra],
    qa bf16 [..., Nq, 96 + extra], scale) -- the product is not stored, qa's rel-pos columns are written."""
    _chk_rows(a)
    _chk_dev(w, bias, out, out2, aux, row_scale)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.dtype == BF16 and w.dtype == BF16
    if epilogue == hip.EPI_RELQ:
        cmap, qa, scale = relq
        _chk_dev(cmap, qa)
        extra = qa.shape[-1] - 96
        assert cmap.dtype == torch.int32 and cmap.is_contiguous() and cmap.shape == (qa.shape[-2], extra)
        assert qa.is_contiguous() and qa.numel() == M * qa.shape[-1] and bias is None
        g = hip.GemmArgs()
        g.A, g.lda, g.W, g.ldw = ptr(a), a.stride(0), ptr(w), w.stride(0)
        g.M, g.N, g.K, g.epilogue = M, N, K, epilogue
        g.relq_map, g.relq_out, g.relq_ld = ptr(cmap), ptr(qa), qa.shape[-1]
        g.relq_extra, g.relq_rows, g.relq_scale = extra, qa.shape[-2], scale
        hip.call("svit_gemm_nt", C.byref(g), meta=("mnk", M, N, K, epilogue, 2 * M * K + 2 * N * K + 2 * M * extra))
        return qa
    if out is None:
        dt = F32 if epilogue in (hip.EPI_RESID, hip.EPI_F32) else BF16
        out = torch.empty((M, N), device=a.device, dtype=dt)
    if epilogue == hip.EPI_GELU and out2 is None and save:
        out2 = torch.empty((M, N), device=a.device, dtype=BF16)
    g = hip.GemmArgs()
    g.A, g.lda, g.W, g.ldw = ptr(a), a.stride(0), ptr(w), w.stride(0)
    g.bias = ptr(bias)
    g.out, g.ldo = ptr(out), out.stride(-2)
    g.out2, g.ldo2 = ptr(out2), (out2.stride(-2) if out2 is not None else 0)
    g.aux, g.ldaux = ptr(aux), (aux.stride(-2) if aux is not None else 0)
    g.row_scale, g.rows_per_sample = ptr(row_scale), rows_per_sample
    g.M, g.N, g.K = M, N, K
    g.epilogue, g.accumulate = epilogue, int(accumulate)
    if remap is not None:
        g.remap_L, g.remap_N, g.remap_off = remap
    # algorithmic HBM bytes of the call (operands once, outputs once; residual / saved-derivative
    # slabs the epilogue reads): what bench.py prices against the 8 TB/s roof
    osz = out.element_size()
    nbytes = 2 * M * K + 2 * N * K + M * N * osz
    if out2 is not None:
        nbytes += 2 * M * N
    if aux is not None:
        nbytes += M * N * aux.element_size()
    if accumulate:
        nbytes += M * N * osz
    hip.call("svit_gemm_nt", C.byref(g), meta=("mnk", M, N, K, epilogue, nbytes))
    return (out, out2) if epilogue == hip.EPI_GELU else out


def gemm_tn(a, b, dw, splits=0, dbias=None):
    """dw[N,K] (f32) += a[M,N]^T @ b[M,K].  a / b may be column slices (row-strided views)."""
    _chk_rows(a, b)
    _chk_dev(dw)
    M, N = a.shape
    K = dw.shape[-1]  # may be smaller than b's padded width (patch-embed wgrad)
    assert b.shape[0] == M and b.shape[1] >= K and dw.shape[-2] == N and dw.dtype == F32
    hip.call("svit_gemm_tn", ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(dw), dw.stride(-2),
             M, N, K, splits, ptr(dbias), meta=("mnk", M, N, K))
    return dw


def gemm_tn_grouped(problems, ordered=False):
    """problems: [(a[M,N], b[M,K'], dw[N,K] f32, dbias or None)], each as for gemm_tn; all are
    accumulated by one launch (per group of 8).  ordered: no split along the reduction rows
    (bit-reproducible weight gradients, slower)."""
    n = len(problems)
    if n == 0:
        return
    arr = (hip.TnProblem * n)()
    flop = 0.0
    for i, (a, b, dw, dbias) in enumerate(problems):
        _chk_rows(a, b)
        _chk_dev(dw)
        M, N = a.shape
        K = dw.shape[-1]
        assert b.shape[0] == M and b.shape[1] >= K and dw.shape[-2] == N and dw.dtype == F32
        t = arr[i]
        t.A, t.B, t.dW, t.dbias = ptr(a), ptr(b), ptr(dw), ptr(dbias)
        t.lda, t.ldb, t.lddw, t.M, t.N, t.K = a.stride(0), b.stride(0), dw.stride(-2), M, N, K
        flop += 2.0 * M * N * K
    if ordered:
        hip.call("svit_gemm_tn_grouped_ex", arr, n, 1, meta=("flop", flop))
    else:
        hip.call("svit_gemm_tn_grouped", arr, n, meta=("flop", flop))


def reduce_defer(on):
    """queue (1) / run (0) the second-stage reduce launches of the backward kernels"""
    hip.call("svit_reduce_defer", int(on))


def reduce_flush():
    """run the current stream's queued second-stage reductions now (the stream stays in deferred mode)"""
    hip.call("svit_reduce_flush")


def reduce_reset():
    """error path: drop the current stream's queued reductions and leave deferred mode"""
    hip.call("svit_reduce_reset")


def colsum(a, out):
    _chk_dev(a, out)
    hip.call("svit_colsum_bf16", ptr(a), a.stride(0), ptr(out), a.shape[0], a.shape[1])
    return out


def cast_bf16(src, dst=None):
    _chk_dev(src, dst)
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=BF16)
    hip.call("svit_cast_f32_bf16", ptr(src), ptr(dst), src.numel())
    return dst


def transpose_cast_batched(src_flat, dst_flat, table, n_mats, max_tiles):
    hip.call("svit_transpose_cast_batched", ptr(src_flat), ptr(dst_flat), ptr(table), n_mats,
             max_tiles)


def transpose_bf16_batched(src16_flat, dst_flat, table, n_mats, max_tiles):
    """the same table of [R,C] -> [C,R] transposes, read from the bf16 mirror (offsets index the mirror)"""
    _chk_dev(src16_flat, dst_flat, table)
    assert src16_flat.dtype == BF16 and dst_flat.dtype == BF16
    hip.call("svit_transpose_bf16_batched", ptr(src16_flat), ptr(dst_flat), ptr(table), n_mats,
             max_tiles)


def table_interp(mcat, tables, want_f32=True):
    """rel-pos tables at another resolution: (mcat f32 [Lp, J]) @ (tables f32 [J, 96]) -> (f32 [Lp, 96] or None, bf16 [Lp, 96])
    in one launch."""
    _chk_dev(mcat, tables)
    assert mcat.dtype == F32 and tables.dtype == F32 and mcat.is_contiguous() and tables.is_contiguous()
    assert mcat.shape[1] == tables.shape[0] and tables.shape[1] == HD
    r32 = torch.empty((mcat.shape[0], HD), device=mcat.device, dtype=F32) if want_f32 else None
    r16 = torch.empty((mcat.shape[0], HD), device=mcat.device, dtype=BF16)
    hip.call("svit_table_interp", ptr(mcat), mcat.shape[0], mcat.shape[1], ptr(tables), ptr(r32), ptr(r16))
    return r32, r16


def table_interp_jobs(entries, device):
    """entries: [(mcat f32 [Lp, J], tables f32 [J, 96], out32 f32 [Lp, 96] or None, out16 bf16 [Lp, 96])] -> the device
    descriptor table of svit_table_interp_batched (uint8 [n, 40]; keep it and every tensor it names alive)."""
    import struct
    raw = b"".join(struct.pack("<QQQQii", ptr(m), ptr(t), ptr(o32) or 0, ptr(o16) or 0, m.shape[0], m.shape[1])
                   for m, t, o32, o16 in entries)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).view(len(entries), 40).to(device)


def table_interp_batched(jobs, max_rows):
    hip.call("svit_table_interp_batched", ptr(jobs), jobs.shape[0], max_rows)


def pad_cast_rows(src, dst):
    """dst bf16 [R,ldd] = [src f32 [R,C] | 0]."""
    _chk_dev(src, dst)
    hip.call("svit_pad_cast_rows", ptr(src), ptr(dst), src.shape[0], src.shape[1], dst.shape[1])
    return dst


def scale_cast(src, row_scale=None, rows_per_sample=0, dst=None, gather=None):
    """bf16(row_scale * src).  gather=(L, off): src is [B, N, C] and only rows [off, off+L) of
    every sample are converted -> dst [B*L, C]."""
    _chk_dev(src, row_scale, dst)
    cols = src.shape[-1]
    if gather is None:
        rows, gl, gn, go = src.shape[:-1].numel(), 0, 0, 0
        shape = src.shape
    else:
        gl, go = gather
        gn = src.shape[-2]
        rows = src.shape[0] * gl
        shape = (rows, cols)
    if dst is None:
        dst = torch.empty(shape, device=src.device, dtype=BF16)
    hip.call("svit_scale_cast", ptr(src), ptr(dst), ptr(row_scale), rows_per_sample, rows, cols,
             gl, gn, go)
    return dst


def layernorm_fwd(x, gamma, beta, eps=1e-6, want_f32=False, want_bf16=True, save_stats=True):
    _chk_dev(x, gamma, beta)
    C_ = x.shape[-1]
    rows = x.numel() // C_
    y16 = torch.empty(x.shape, device=x.device, dtype=BF16) if want_bf16 else None
    y32 = torch.empty(x.shape, device=x.device, dtype=F32) if want_f32 else None
    mean = torch.empty(rows, device=x.device, dtype=F32) if save_stats else None
    rstd = torch.empty(rows, device=x.device, dtype=F32) if save_stats else None
    hip.call("svit_layernorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(y16), ptr(y32), ptr(mean),
             ptr(rstd), rows, C_, eps)
    return y16, y32, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=None, dx=None, want_bf16=False,
                  row_scale=None, rows_per_sample=0, ws=None):
    """-> dx f32, or (dx, bf16(row_scale * dx)) with want_bf16 (operand of the next GEMM)."""
    _chk_dev(dy, x, gamma, mean, rstd, dgamma, dbeta, dres, dx, row_scale)
    C_ = x.shape[-1]
    rows = x.numel() // C_
    if dx is None:
        dx = torch.empty(x.shape, device=x.device, dtype=F32)
    dx16 = torch.empty(x.shape, device=x.device, dtype=BF16) if want_bf16 else None
    ws = scratch(x.device) if ws is None else ws
    assert dy.dtype in (F32, BF16)
    hip.call("svit_layernorm_bwd", ptr(dy), int(dy.dtype == BF16), ptr(x), ptr(gamma), ptr(mean),
             ptr(rstd), ptr(dres),
             ptr(dx), ptr(dx16), ptr(row_scale), rows_per_sample, ptr(dgamma), ptr(dbeta), rows, C_,
             ptr(ws), ws.numel())
    return (dx, dx16) if want_bf16 else dx


def im2col_patch(video):
    _chk_dev(video)
    B, Cin, T, H, W = video.shape
    assert Cin == 3 and video.dtype == F32
    To, Ho, Wo = (T - 1) // 2 + 1, (H - 1) // 4 + 1, (W - 1) // 4 + 1
    cols = torch.empty((B * To * Ho * Wo, 448), device=video.device, dtype=BF16)
    hip.call("svit_im2col_patch", ptr(video), ptr(cols), B, T, H, W)
    return cols, (To, Ho, Wo)


def im2col_patch_u8(clips):
    """svit_amd.input.U8Clips -> the same [rows, 448] bf16 operand as im2col_patch."""
    fr = clips.frames
    _chk_dev(fr)
    V, T, Hs, Ws, _ = fr.shape
    B, S = clips.crops.shape[0], clips.size
    To, Ho, Wo = (T - 1) // 2 + 1, (S - 1) // 4 + 1, (S - 1) // 4 + 1
    cols = torch.empty((B * To * Ho * Wo, 448), device=fr.device, dtype=BF16)
    hip.call("svit_im2col_patch_u8", ptr(fr), fr.numel(), ptr(clips.lut), ptr(clips.crops),
             ptr(cols), B, T, Hs, Ws, S)
    return cols, (To, Ho, Wo)


def fill_special_tokens(x, cls, objq, pos_t, L, Tx, O, add_pos):
    B, N, C_ = x.shape
    hip.call("svit_fill_special_tokens", ptr(x), ptr(cls), ptr(objq), ptr(pos_t), B, N, L, Tx, O,
             C_, int(add_pos))


def special_token_grads(dx, g_cls, g_obj, g_pos, L, Tx, O, add_pos):
    """gradients of cls_token / object_queries / pos_embed_temporal from d(block-0 input), ACCUMULATED, one launch"""
    B, N, C_ = dx.shape
    assert dx.is_contiguous() and dx.dtype == F32
    hip.call("svit_special_token_grads", ptr(dx), ptr(g_cls), ptr(g_obj), ptr(g_pos), B, N, L, Tx, O, C_, int(add_pos))


def pooled(n, s):
    return (n - 1) // s + 1


# no-grad passes over clips (T > 1: eval, multi-view test) hand the kernels a scratch pre / mean / rstd so that the pooling
# forward takes the same staged / slab launches as a training step (they hand `pre` from the conv launch to the LayerNorm
# launch); one-plane volumes (the frames pass) have their own kernel and keep nothing
NOGRAD_SCRATCH = True


def _pool_fwd_args(a, qkv, which, conv_w, gamma, beta, B, heads, thw, n_obj, stride_hw, ld_out, mode,
                   eps, save=True, out_scale=1.0):
    _chk_dev(qkv, conv_w, gamma, beta)
    T, H, W = thw
    Nout = 1 + T * pooled(H, stride_hw) * pooled(W, stride_hw) + n_obj
    dev = qkv.device
    out = torch.empty((B, heads, Nout, ld_out), device=dev, dtype=BF16)
    keep = save or (NOGRAD_SCRATCH and T > 1)
    pre = torch.empty((B, heads, Nout, HD), device=dev, dtype=BF16) if keep else None
    mean = torch.empty(B * heads * Nout, device=dev, dtype=F32) if keep else None
    rstd = torch.empty(B * heads * Nout, device=dev, dtype=F32) if keep else None
    a.qkv, a.which, a.conv_w, a.gamma, a.beta = ptr(qkv), which, ptr(conv_w), ptr(gamma), ptr(beta)
    a.out, a.ld_out, a.pre, a.mean, a.rstd = ptr(out), ld_out, ptr(pre), ptr(mean), ptr(rstd)
    a.B, a.heads, a.T, a.H, a.W, a.n_obj = B, heads, T, H, W, n_obj
    a.stride_hw, a.mode, a.eps, a.out_scale = stride_hw, mode, eps, out_scale
    return out, pre, mean, rstd       # (scratch included: the caller drops it AFTER the launch -- freed earlier, the next
                                      # tensor's `out` could be carved from the same block)


def pool_ln_fwd(qkv, which, conv_w, gamma, beta, B, heads, thw, n_obj, stride_hw, ld_out=HD,
                mode=0, eps=1e-6, out_scale=1.0):
    """-> out bf16 [B,h,Nout,ld_out], pre bf16 [B,h,Nout,96], mean, rstd f32 [B*h*Nout]."""
    a = hip.PoolArgs()
    res = _pool_fwd_args(a, qkv, which, conv_w, gamma, beta, B, heads, thw, n_obj, stride_hw,
                         ld_out, mode, eps, out_scale=out_scale)
    hip.call("svit_pool_ln_fwd", C.byref(a))
    return res


def pool_weight_sel(src_flat, offsets, dst):
    """selector tables (uint32 [n, 27, 96]) of n depthwise weights living at `offsets` (int64
    element offsets) of the fp32 buffer `src_flat` -- scalar operands of the tiled stencils."""
    _chk_dev(src_flat, offsets, dst)
    hip.call("svit_pool_weight_sel", ptr(src_flat), ptr(offsets), ptr(dst), offsets.numel())
    return dst


def _sel_ptrs(sels):
    arr = (C.c_void_p * 3)()
    for i in range(3):
        arr[i] = ptr(sels[i])
    return arr


def pool_ln_fwd_qkv(qkv, conv_ws, gammas, betas, B, heads, thw, n_obj, strides, ld_outs, modes,
                    eps=1e-6, save=True, sels=None, out_scales=(1.0, 1.0, 1.0), relq=None):
    """q, k, v pooling + LayerNorm in one launch -> [(out, pre, mean, rstd)] * 3 (the last three
    are None with save=False: no-grad passes keep nothing for a backward).  sels: the three
    selector tables (pool_weight_sel) -- stride-1 tensors then run the LDS-tiled stencil.
    relq = (rcat bf16 [Lpad, 96], map i32 [Nq, ld_out - 96], scale): the q tensor's rel-pos columns
    out[..., 96:] are written too (inside the slab LayerNorm kernel, or by a GEMM the entry point adds)."""
    arr = (hip.PoolArgs * 3)()
    res = [_pool_fwd_args(arr[i], qkv, i, conv_ws[i], gammas[i], betas[i], B, heads, thw, n_obj,
                          strides[i], ld_outs[i], modes[i], eps, save, out_scale=out_scales[i])
           for i in range(3)]
    if relq is not None:
        rcat, cmap, rscale = relq
        _chk_dev(rcat, cmap)
        assert rcat.dtype == BF16 and rcat.is_contiguous() and rcat.shape[1] == HD
        assert cmap.dtype == torch.int32 and cmap.is_contiguous()
        assert tuple(cmap.shape) == (res[0][0].shape[2], ld_outs[0] - HD)
        arr[0].relq_R, arr[0].relq_map, arr[0].relq_lpad, arr[0].relq_scale = ptr(rcat), ptr(cmap), rcat.shape[0], rscale
    if sels is None:
        hip.call("svit_pool_ln_fwd_qkv", arr)
    else:
        hip.call("svit_pool_ln_fwd_qkv_sel", arr, _sel_ptrs(sels))
    if not save:
        res = [(r[0], None, None, None) for r in res]
    return res


def _pool_ln_bwd_args(a, pre, mean, rstd, gamma, dgamma, dbeta, B, heads, Nout, d_main=None,
                      ld_main=HD, d_res=None, d_extra=None, ws=None):
    dpre = torch.empty((B, heads, Nout, HD), device=pre.device, dtype=BF16)
    a.d_main = ptr(d_main)
    a.main_is_f32 = int(d_main is not None and d_main.dtype == F32)
    a.ld_main = ld_main
    a.main_parts, a.main_part_stride = 1, 0
    if d_main is not None and d_main.dim() == 5:        # [parts, B, h, N, 96]: attn_bwd's partial planes
        assert d_main.dtype == F32
        a.main_parts, a.main_part_stride = d_main.shape[0], d_main.stride(0)
    a.d_res, a.d_extra = ptr(d_res), ptr(d_extra)
    a.extra_is_bf16 = int(d_extra is not None and d_extra.dtype == BF16)
    a.pre, a.mean, a.rstd, a.gamma = ptr(pre), ptr(mean), ptr(rstd), ptr(gamma)
    a.dpre, a.dgamma, a.dbeta = ptr(dpre), ptr(dgamma), ptr(dbeta)
    a.B, a.heads, a.Nout = B, heads, Nout
    ws = scratch(pre.device) if ws is None else ws
    a.workspace, a.workspace_floats = ptr(ws), ws.numel()
    return dpre


def pool_ln_bwd(pre, mean, rstd, gamma, dgamma, dbeta, B, heads, Nout, d_main=None, ld_main=HD,
                d_res=None, d_extra=None):
    a = hip.PoolLnBwdArgs()
    dpre = _pool_ln_bwd_args(a, pre, mean, rstd, gamma, dgamma, dbeta, B, heads, Nout, d_main,
                             ld_main, d_res, d_extra)
    hip.call("svit_pool_ln_bwd", C.byref(a))
    return dpre


def pool_ln_bwd_qkv(entries, ws=None):
    """entries: 3 x (args tuple, kwargs dict) of pool_ln_bwd -> [dpre] * 3, one launch."""
    arr = (hip.PoolLnBwdArgs * 3)()
    res = [_pool_ln_bwd_args(arr[i], *entries[i][0], ws=ws, **entries[i][1]) for i in range(3)]
    hip.call("svit_pool_ln_bwd_qkv", arr)
    return res


# (the four streaming conv-backward entry points left the product library in round 6: their wrappers live in
#  tools/diag/pool_streaming.py and need a -DSVIT_DIAG_POOL_STREAMING build)
def _pool_dgrad_args(a, dpre, conv_w, dqkv, which, B, heads, thw, n_obj, stride_hw):
    a.dpre, a.conv_w, a.dqkv, a.which = ptr(dpre), ptr(conv_w), ptr(dqkv), which
    a.B, a.heads, (a.T, a.H, a.W), a.n_obj, a.stride_hw = B, heads, thw, n_obj, stride_hw




def _pool_wgrad_args(a, dpre, qkv, which, dw, B, heads, thw, n_obj, stride_hw, ws=None):
    a.dpre, a.qkv, a.which, a.dw = ptr(dpre), ptr(qkv), which, ptr(dw)
    a.B, a.heads, (a.T, a.H, a.W), a.n_obj, a.stride_hw = B, heads, thw, n_obj, stride_hw
    ws = scratch(dpre.device) if ws is None else ws
    a.workspace, a.workspace_floats = ptr(ws), ws.numel()




def pool_conv_bwd_workspace(B, heads, thw, n_obj, strides):
    """floats of workspace pool_conv_bwd_qkv needs for this geometry (grows with the batch); -1 = no plan."""
    da = (hip.PoolDgradArgs * 3)()
    for i in range(3):
        da[i].B, da[i].heads, (da[i].T, da[i].H, da[i].W), da[i].n_obj, da[i].stride_hw, da[i].which = B, heads, thw, n_obj, strides[i], i
    return int(hip.load().svit_pool_conv_bwd_workspace(da))


def pool_conv_bwd_qkv(dpres, conv_ws, dqkv, qkv, dws, B, heads, thw, n_obj, strides, ws=None):
    """conv dgrad + conv wgrad of q, k, v: one fused launch (csrc/pool.hip::pool_bwd_fused_kernel).  ws: the workspace of its
    partial rows (>= pool_conv_bwd_workspace(...) floats; it must outlive the deferred second-stage reduce); None = a per-device
    scratch of the needed size."""
    if ws is None:
        need = pool_conv_bwd_workspace(B, heads, thw, n_obj, strides)
        ws = scratch(dpres[0].device, max(8 * 1024 * 1024, need), tag="poolbwd")
    da = (hip.PoolDgradArgs * 3)()
    wa = (hip.PoolWgradArgs * 3)()
    for i in range(3):
        _pool_dgrad_args(da[i], dpres[i], conv_ws[i], dqkv, i, B, heads, thw, n_obj, strides[i])
        _pool_wgrad_args(wa[i], dpres[i], qkv, i, dws[i], B, heads, thw, n_obj, strides[i], ws)
    hip.call("svit_pool_conv_bwd_qkv", da, wa)


def relpos_q_fwd(qa, tabs, idx, B, heads, q_thw, k_thw, n_obj, inv_scale):
    a = hip.RelqArgs()
    a.qa, a.ld = ptr(qa), qa.shape[-1]
    a.rel_h, a.rel_w, a.rel_t = (ptr(t) for t in tabs)
    a.idx_h, a.idx_w, a.idx_t = (ptr(t) for t in idx)
    a.B, a.heads = B, heads
    a.qt, a.qh, a.qw = q_thw
    a.kt, a.kh, a.kw = k_thw
    a.n_obj, a.inv_scale = n_obj, inv_scale
    hip.call("svit_relpos_q_fwd", C.byref(a))


def relpos_q_bwd(qa, dqa, tabs, idx, dtabs, B, heads, q_thw, k_thw, n_obj, inv_scale):
    Nq = qa.shape[2]
    dq_extra = torch.empty((B, heads, Nq, HD), device=qa.device, dtype=F32)
    a = hip.RelqBwdArgs()
    a.qa, a.dqa, a.ld = ptr(qa), ptr(dqa), qa.shape[-1]
    a.rel_h, a.rel_w, a.rel_t = (ptr(t) for t in tabs)
    a.idx_h, a.idx_w, a.idx_t = (ptr(t) for t in idx)
    a.dq_extra = ptr(dq_extra)
    a.drel_h, a.drel_w, a.drel_t = (ptr(t) for t in dtabs)
    a.rows_h, a.rows_w, a.rows_t = (t.shape[0] for t in tabs)
    a.B, a.heads = B, heads
    a.qt, a.qh, a.qw = q_thw
    a.kt, a.kh, a.kw = k_thw
    a.n_obj, a.inv_scale = n_obj, inv_scale
    ws = scratch(qa.device)
    a.workspace, a.workspace_floats = ptr(ws), ws.numel()
    hip.call("svit_relpos_q_bwd", C.byref(a))
    return dq_extra


def relpos_gather(P, qa, idx, rows_off, B, heads, q_thw, k_thw, n_obj, inv_scale):
    """qa[..., 96 + j] = P[token, rows_off[sec] + idx] * inv_scale (see svit_relpos_gather)."""
    a = hip.RelqGatherArgs()
    a.P, a.ldp, a.qa, a.ld = ptr(P), P.shape[-1], ptr(qa), qa.shape[-1]
    a.idx_h, a.idx_w, a.idx_t = (ptr(t) for t in idx)
    a.row_h, a.row_w, a.row_t = rows_off
    a.B, a.heads = B, heads
    a.qt, a.qh, a.qw = q_thw
    a.kt, a.kh, a.kw = k_thw
    a.n_obj, a.inv_scale = n_obj, inv_scale
    hip.call("svit_relpos_gather", C.byref(a))


def relpos_scatter(dqa, idx, offs, ldd, B, heads, q_thw, k_thw, n_obj, inv_scale):
    """-> D bf16 [B*h*Nq, ldd] with d(relq)*inv_scale scattered to table-row columns."""
    Nq = dqa.shape[2]
    D = torch.empty((B * heads * Nq, ldd), device=dqa.device, dtype=BF16)
    a = hip.RelqScatterArgs()
    a.dqa, a.ld, a.D, a.ldd = ptr(dqa), dqa.shape[-1], ptr(D), ldd
    a.idx_h, a.idx_w, a.idx_t = (ptr(t) for t in idx)
    a.off_h, a.off_w, a.off_t = offs
    a.B, a.heads = B, heads
    a.qt, a.qh, a.qw = q_thw
    a.kt, a.kh, a.kw = k_thw
    a.n_obj, a.inv_scale = n_obj, inv_scale
    hip.call("svit_relpos_scatter", C.byref(a))
    return D


def attn_fwd(qa, ka, v, scale, bias_cols=0):
    """qa [B,h,Nq,DA], ka [B,h,Nk,DA], v [B,h,Nk,96] -> ctx bf16 [B,Nq,h*96], lse2 [B,h,Nq].
    bias_cols = kt + kh + kw (rel-pos columns that carry data; 0 = all DA - 96)."""
    _chk_dev(qa, ka, v)
    B, heads, Nq, DA = qa.shape
    Nk = ka.shape[2]
    ctx = torch.empty((B, Nq, heads * HD), device=qa.device, dtype=BF16)
    lse2 = torch.empty((B, heads, Nq), device=qa.device, dtype=F32)
    a = hip.AttnFwdArgs()
    a.qa, a.ka, a.v, a.ctx, a.lse2 = ptr(qa), ptr(ka), ptr(v), ptr(ctx), ptr(lse2)
    a.B, a.heads, a.Nq, a.Nk, a.DA, a.scale, a.bias_cols = B, heads, Nq, Nk, DA, scale, bias_cols
    hip.call("svit_attn_fwd", C.byref(a), meta=("attn", B, heads, Nq, Nk, DA))
    return ctx, lse2


def attn_bwd(qa, ka, v, ctx, dctx, lse2, scale, q_splits=0, bias_cols=0, reld=None):
    """-> dqa bf16 [B,h,Nq,DA], dk f32 [parts,B,h,Nk,96], dv f32 [parts,B,h,Nk,96]: the gradients of k
    and v are the SUMS over the leading axis (one plane per chunk of the query range; pool_ln_bwd
    adds them while it reads).  reld = (map i32 [Nq, DA - 96], ldd, scale[, rt]): the dq kernel also writes the
    rel-pos backward's scattered matrix D bf16 [B*h*Nq, ldd] (what relpos_scatter builds), returned 4th, and --
    given rt = the transposed tables bf16 [96, ldd], ldd <= 128 -- dq_extra = D . rt^T f32 [B*h*Nq, 96], returned
    5th (None when it was not computed: the caller then runs the GEMM); with a fifth element "fold" the product
    is added into dqa[..., :96] inside the kernel instead and the 5th result is the string "folded"."""
    _chk_dev(qa, ka, v, ctx, dctx, lse2)
    B, heads, Nq, DA = qa.shape
    Nk = ka.shape[2]
    dev = qa.device
    a = hip.AttnBwdArgs()
    a.B, a.heads, a.Nq, a.Nk, a.DA, a.q_splits, a.scale = B, heads, Nq, Nk, DA, q_splits, scale
    a.bias_cols = bias_cols
    parts = hip.load().svit_attn_bwd_parts(C.byref(a))
    if parts < 1:
        raise hip.SvitHipError("svit_attn_bwd_parts failed: %d" % parts)
    dqa = torch.empty((B, heads, Nq, DA), device=dev, dtype=BF16)
    dkv = torch.empty((2, parts, B, heads, Nk, HD), device=dev, dtype=F32)
    delta = torch.empty((B, heads, Nq, 2), device=dev, dtype=F32)
    a.qa, a.ka, a.v, a.ctx, a.dctx, a.lse2 = (ptr(t) for t in (qa, ka, v, ctx, dctx, lse2))
    a.delta, a.dqa, a.dk, a.dv = ptr(delta), ptr(dqa), ptr(dkv[0]), ptr(dkv[1])
    a.q_splits = parts
    D = X = None
    if reld is not None:
        cmap, ldd, rscale = reld[:3]
        rt = reld[3] if len(reld) > 3 else None      # bf16 [96, ldd]: also multiply dq_extra = D . rt^T
        _chk_dev(cmap, rt)
        assert cmap.dtype == torch.int32 and cmap.is_contiguous() and cmap.shape == (Nq, DA - HD)
        D = torch.empty((B * heads * Nq, ldd), device=dev, dtype=BF16)
        a.relD, a.relD_ld, a.relD_map, a.relD_scale = ptr(D), ldd, ptr(cmap), rscale
        mode = reld[4] if len(reld) > 4 else "separate"
        if rt is not None and ldd <= 128 and ldd % 16 == 0:
            assert rt.dtype == BF16 and rt.is_contiguous() and tuple(rt.shape) == (HD, ldd)
            a.relR = ptr(rt)
            if mode == "fold":        # dqa[:, :96] += D . rt^T inside the kernel: no dq_extra tensor at all
                X = "folded"
            else:
                X = torch.empty((B * heads * Nq, HD), device=dev, dtype=F32)
                a.relX = ptr(X)
    # (the D . R^T product of the rel-pos backward, when this launch carries it: 2 * rows * 96 * ldd flops that the
    # step used to spend in a GEMM launch of its own -- bench.py adds them to this kernel's algorithmic count)
    folded = 2.0 * B * heads * Nq * HD * reld[1] if (reld is not None and X is not None) else 0.0
    hip.call("svit_attn_bwd", C.byref(a), meta=("attn", B, heads, Nq, Nk, DA, folded))
    if reld is not None:
        return dqa, dkv[0], dkv[1], D, X
    return dqa, dkv[0], dkv[1]


def maxpool_fwd(x, thw, n_obj):
    B, N, C_ = x.shape
    T, H, W = thw
    Nout = 1 + T * pooled(H, 2) * pooled(W, 2) + n_obj
    y = torch.empty((B, Nout, C_), device=x.device, dtype=F32)
    idx = torch.empty((B, Nout, C_), device=x.device, dtype=torch.uint8)
    hip.call("svit_maxpool_fwd", ptr(x), ptr(y), ptr(idx), B, T, H, W, n_obj, C_)
    return y, idx


def maxpool_bwd(dy, idx, thw, n_obj, bf16=False):
    """bf16=True: dx rounded to bf16 (= scale_cast(maxpool_bwd(...)) bit for bit, one launch)."""
    B, Nout, C_ = dy.shape
    T, H, W = thw
    dx = torch.empty((B, 1 + T * H * W + n_obj, C_), device=dy.device, dtype=BF16 if bf16 else F32)
    hip.call("svit_maxpool_bwd_bf16" if bf16 else "svit_maxpool_bwd", ptr(dy), ptr(idx), ptr(dx),
             B, T, H, W, n_obj, C_)
    return dx


def sumsq(g, out):
    ws = scratch(g.device)
    hip.call("svit_sumsq", ptr(g), g.numel(), ptr(out), ptr(ws), ws.numel())


def adamw_step(p, g, m, v, sumsq_t, max_norm, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    hip.call("svit_adamw_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(sumsq_t), max_norm,
             lr, beta1, beta2, eps, wd, step, grad_scale)


# ---------------------------------------------------------------- head (K15) ----------------
def _head_args(tokens, T, O, keep, head_params, outs):
    a = hip.HeadArgs()
    B, N, C_ = tokens.shape
    a.tokens, a.B, a.N, a.C, a.T, a.O = ptr(tokens), B, N, C_, T, O
    a.keep = ptr(keep)
    (wp, bp), (wb, bb), (we, be), (wc, bc) = head_params
    a.w_proj, a.b_proj, a.n_cls = ptr(wp), ptr(bp), wp.shape[0]
    a.w_box, a.b_box, a.w_bce, a.b_bce, a.w_con, a.b_con = ptr(wb), ptr(bb), ptr(we), ptr(be), ptr(wc), ptr(bc)
    a.logits, a.boxes, a.contact, a.xobj = (ptr(t) for t in outs)
    return a


def head_fwd(tokens, T, O, keep, head_params):
    """SViT head in one launch (svit_head_fwd): tokens f32 [B,N,C] -> logits [B,n_cls], pred_bboxes [B,T,O,5],
    contact [B,T,2,5], obj_desc [B,T,O,C].  head_params = ((w,b) of projection, box MLP, objectness, contact)."""
    assert tokens.is_contiguous() and tokens.dtype == F32
    _chk_dev(tokens, keep, *[t for wb in head_params for t in wb])
    B, N, C_ = tokens.shape
    dev = tokens.device
    outs = (torch.empty((B, head_params[0][0].shape[0]), device=dev), torch.empty((B, T, O, 5), device=dev),
            torch.empty((B, T, 2, 5), device=dev), torch.empty((B, T, O, C_), device=dev))
    a = _head_args(tokens, T, O, keep, head_params, outs)
    hip.call("svit_head_fwd", C.byref(a))
    return outs


def head_bwd(tokens, T, O, keep, head_params, boxes, grads_out, param_grads):
    """-> d(tokens) f32 [B,N,C] (zero rows for the patch tokens); the parameter gradients are ADDED into
    param_grads (same nesting as head_params).  grads_out = (dlogits, dboxes, dcontact, dxobj), None allowed."""
    assert tokens.is_contiguous() and tokens.dtype == F32
    go = [None if t is None else t.contiguous() for t in grads_out]
    _chk_dev(tokens, keep, boxes, *[t for t in go if t is not None])
    g = hip.HeadBwdArgs()
    # of the forward outputs only the sigmoid boxes are read; the argument check wants the three pointers set
    g.f = _head_args(tokens, T, O, keep, head_params, (boxes, boxes, boxes, None))
    g.dlogits, g.dboxes, g.dcontact, g.dxobj = (ptr(t) for t in go)
    dtok = torch.empty_like(tokens)
    g.dtokens = ptr(dtok)
    (gwp, gbp), (gwb, gbb), (gwe, gbe), (gwc, gbc) = param_grads
    g.gw_proj, g.gb_proj, g.gw_box, g.gb_box = ptr(gwp), ptr(gbp), ptr(gwb), ptr(gbb)
    g.gw_bce, g.gb_bce, g.gw_con, g.gb_con = ptr(gwe), ptr(gbe), ptr(gwc), ptr(gbc)
    hip.call("svit_head_bwd", C.byref(g))
    return dtok


# ---------------------------------------------------------------- image-rank HAOG losses ----
def haog_loss_fwd(pred, tar, contact, contact_tar):
    """-> (losses f32 [8], (g_l1, g_bce, g_giou, g_contact)); see include/svit_hip.h."""
    R, Rc = pred.numel() // 5, contact.numel() // 5
    assert pred.dtype == F32 and tar.dtype == F32 and contact.dtype == F32
    assert contact_tar.dtype == torch.int64 and tar.numel() == R * 4 and contact_tar.numel() == Rc
    dev = pred.device
    losses = torch.empty(8, dtype=F32, device=dev)
    g_l1, g_giou = torch.empty((R, 4), dtype=F32, device=dev), torch.empty((R, 4), dtype=F32, device=dev)
    g_bce, g_contact = torch.empty(R, dtype=F32, device=dev), torch.empty((Rc, 5), dtype=F32, device=dev)
    hip.call("svit_haog_loss", ptr(pred), ptr(tar), ptr(contact), ptr(contact_tar), ptr(losses),
             ptr(g_l1), ptr(g_bce), ptr(g_giou), ptr(g_contact), R, Rc)
    return losses, (g_l1, g_bce, g_giou, g_contact)


def haog_loss_bwd(upstream, unit, pred_shape, contact_shape):
    g_l1, g_bce, g_giou, g_contact = unit
    R, Rc = g_bce.numel(), g_contact.numel() // 5
    dpred = torch.empty(pred_shape, dtype=F32, device=g_bce.device)
    dcontact = torch.empty(contact_shape, dtype=F32, device=g_bce.device)
    hip.call("svit_haog_loss_bwd", ptr(upstream), ptr(g_l1), ptr(g_bce), ptr(g_giou),
             ptr(g_contact), ptr(dpred), ptr(dcontact), R, Rc)
    return dpred, dcontact


def ce_loss(logits, labels):
    """mean cross entropy over the rows (ignore_index -100) and d loss / d logits in one launch (svit_ce_loss)."""
    _chk_dev(logits, labels)
    assert logits.dtype == F32 and logits.dim() == 2 and labels.dtype == torch.int64 and labels.numel() == logits.shape[0]
    logits, labels = logits.contiguous(), labels.contiguous()
    loss = torch.empty((), device=logits.device, dtype=F32)
    dlogits = torch.empty_like(logits)
    hip.call("svit_ce_loss", ptr(logits), ptr(labels), logits.shape[0], logits.shape[1], ptr(loss), ptr(dlogits))
    return loss, dlogits


def step_draws(state, keep, per_block, n_drop=0, p_drop=0.0):
    """state int64 [3] = {seed, draw number, 0} (advanced by the launch); keep f32 [n_blocks] -> scales f32 [n_blocks, per_block] =
    floor(keep + U) / keep and drop f32 [n_drop] in {0, 1 / (1 - p)} (svit_step_draws; one launch)."""
    _chk_dev(state, keep)
    assert state.dtype == torch.int64 and state.numel() == 3 and keep.dtype == F32
    nb = keep.numel()
    scales = torch.empty((nb, per_block), device=state.device, dtype=F32)
    drop = torch.empty((n_drop,), device=state.device, dtype=F32) if n_drop else None
    hip.call("svit_step_draws", ptr(state), ptr(keep), nb, per_block, ptr(scales), n_drop, float(p_drop), ptr(drop))
    return scales, drop
