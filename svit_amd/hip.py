"""ctypes binding of libsvit_hip.so (include/svit_hip.h) -- the only door into the HIP kernels.

No torch types cross the C ABI: tensors are passed as raw device pointers + sizes, the stream
as torch's current HIP stream handle.  There is NO fallback: if the library is missing or a
call fails, the caller gets an exception (the product path never routes through PyTorch
reference code or the CPU oracle).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: the library binds to torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVIT_HIP_LIB: a diagnostic build of the same library (tools/diag), never set in production
LIB_PATH = os.environ.get("SVIT_HIP_LIB") or os.path.join(_HERE, "lib", "libsvit_hip.so")
_lib = None

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

EPI_BF16, EPI_GELU, EPI_RESID, EPI_F32, EPI_DGELU, EPI_RELQ = 0, 1, 2, 3, 4, 5


class TnProblem(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("dW", vp), ("dbias", vp), ("lda", i32), ("ldb", i32),
                ("lddw", i32), ("M", i32), ("N", i32), ("K", i32)]


class GemmArgs(C.Structure):
    _fields_ = [("A", vp), ("lda", i32), ("W", vp), ("ldw", i32), ("bias", vp), ("out", vp),
                ("ldo", i32), ("out2", vp), ("ldo2", i32), ("aux", vp), ("ldaux", i32),
                ("row_scale", vp), ("rows_per_sample", i32), ("M", i32), ("N", i32), ("K", i32),
                ("epilogue", i32), ("accumulate", i32), ("remap_L", i32), ("remap_N", i32),
                ("remap_off", i32), ("relq_map", vp), ("relq_out", vp), ("relq_ld", i32),
                ("relq_extra", i32), ("relq_rows", i32), ("relq_scale", f32)]


class PoolArgs(C.Structure):
    _fields_ = [("qkv", vp), ("which", i32), ("conv_w", vp), ("gamma", vp), ("beta", vp),
                ("out", vp), ("ld_out", i32), ("pre", vp), ("mean", vp), ("rstd", vp),
                ("B", i32), ("heads", i32), ("T", i32), ("H", i32), ("W", i32), ("n_obj", i32),
                ("stride_hw", i32), ("mode", i32), ("eps", f32), ("out_scale", f32),
                ("relq_R", vp), ("relq_map", vp), ("relq_lpad", i32), ("relq_scale", f32)]


class PoolLnBwdArgs(C.Structure):
    _fields_ = [("d_main", vp), ("main_is_f32", i32), ("ld_main", i32), ("d_res", vp),
                ("d_extra", vp), ("pre", vp), ("mean", vp), ("rstd", vp), ("gamma", vp),
                ("dpre", vp), ("dgamma", vp), ("dbeta", vp), ("B", i32), ("heads", i32),
                ("Nout", i32), ("workspace", vp), ("workspace_floats", i64), ("main_parts", i32),
                ("main_part_stride", i64), ("extra_is_bf16", i32)]


class PoolDgradArgs(C.Structure):
    _fields_ = [("dpre", vp), ("conv_w", vp), ("dqkv", vp), ("which", i32), ("B", i32),
                ("heads", i32), ("T", i32), ("H", i32), ("W", i32), ("n_obj", i32),
                ("stride_hw", i32)]


class PoolWgradArgs(C.Structure):
    _fields_ = [("dpre", vp), ("qkv", vp), ("which", i32), ("dw", vp), ("B", i32), ("heads", i32),
                ("T", i32), ("H", i32), ("W", i32), ("n_obj", i32), ("stride_hw", i32),
                ("workspace", vp), ("workspace_floats", i64)]


class RelqArgs(C.Structure):
    _fields_ = [("qa", vp), ("ld", i32), ("rel_h", vp), ("rel_w", vp), ("rel_t", vp),
                ("idx_h", vp), ("idx_w", vp), ("idx_t", vp), ("B", i32), ("heads", i32),
                ("qt", i32), ("qh", i32), ("qw", i32), ("kt", i32), ("kh", i32), ("kw", i32),
                ("n_obj", i32), ("inv_scale", f32)]


class RelqBwdArgs(C.Structure):
    _fields_ = [("qa", vp), ("dqa", vp), ("ld", i32), ("rel_h", vp), ("rel_w", vp), ("rel_t", vp),
                ("idx_h", vp), ("idx_w", vp), ("idx_t", vp), ("dq_extra", vp), ("drel_h", vp),
                ("drel_w", vp), ("drel_t", vp), ("rows_h", i32), ("rows_w", i32), ("rows_t", i32),
                ("B", i32), ("heads", i32), ("qt", i32), ("qh", i32), ("qw", i32), ("kt", i32),
                ("kh", i32), ("kw", i32), ("n_obj", i32), ("inv_scale", f32),
                ("workspace", vp), ("workspace_floats", i64)]


class RelqGatherArgs(C.Structure):
    _fields_ = [("P", vp), ("ldp", i32), ("qa", vp), ("ld", i32), ("idx_h", vp), ("idx_w", vp),
                ("idx_t", vp), ("row_h", i32), ("row_w", i32), ("row_t", i32), ("B", i32),
                ("heads", i32), ("qt", i32), ("qh", i32), ("qw", i32), ("kt", i32), ("kh", i32),
                ("kw", i32), ("n_obj", i32), ("inv_scale", f32)]


class RelqScatterArgs(C.Structure):
    _fields_ = [("dqa", vp), ("ld", i32), ("D", vp), ("ldd", i32), ("idx_h", vp), ("idx_w", vp),
                ("idx_t", vp), ("off_h", i32), ("off_w", i32), ("off_t", i32), ("B", i32),
                ("heads", i32), ("qt", i32), ("qh", i32), ("qw", i32), ("kt", i32), ("kh", i32),
                ("kw", i32), ("n_obj", i32), ("inv_scale", f32)]


class AttnFwdArgs(C.Structure):
    _fields_ = [("qa", vp), ("ka", vp), ("v", vp), ("ctx", vp), ("lse2", vp), ("B", i32),
                ("heads", i32), ("Nq", i32), ("Nk", i32), ("DA", i32), ("scale", f32),
                ("bias_cols", i32)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("qa", vp), ("ka", vp), ("v", vp), ("ctx", vp), ("dctx", vp), ("lse2", vp),
                ("delta", vp), ("dqa", vp), ("dk", vp), ("dv", vp), ("B", i32), ("heads", i32),
                ("Nq", i32), ("Nk", i32), ("DA", i32), ("q_splits", i32), ("scale", f32),
                ("bias_cols", i32), ("relD", vp), ("relD_ld", i32), ("relD_map", vp), ("relD_scale", f32),
                ("relR", vp), ("relX", vp)]


class HeadArgs(C.Structure):
    _fields_ = [("tokens", vp), ("B", i32), ("N", i32), ("C", i32), ("T", i32), ("O", i32), ("keep", vp),
                ("w_proj", vp), ("b_proj", vp), ("n_cls", i32), ("w_box", vp), ("b_box", vp),
                ("w_bce", vp), ("b_bce", vp), ("w_con", vp), ("b_con", vp),
                ("logits", vp), ("boxes", vp), ("contact", vp), ("xobj", vp)]


class HeadBwdArgs(C.Structure):
    _fields_ = [("f", HeadArgs), ("dlogits", vp), ("dboxes", vp), ("dcontact", vp), ("dxobj", vp), ("dtokens", vp),
                ("gw_proj", vp), ("gb_proj", vp), ("gw_box", vp), ("gb_box", vp), ("gw_bce", vp), ("gb_bce", vp),
                ("gw_con", vp), ("gb_con", vp)]


_SIGS = {
    "svit_version": (i32, []),
    "svit_arch": (C.c_char_p, []),
    "svit_gemm_nt": (i32, [C.POINTER(GemmArgs), vp]),
    "svit_gemm_tn": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp]),
    "svit_gemm_tn_grouped": (i32, [C.POINTER(TnProblem), i32, vp]),
    "svit_gemm_tn_grouped_ex": (i32, [C.POINTER(TnProblem), i32, i32, vp]),
    "svit_colsum_bf16": (i32, [vp, i32, vp, i32, i32, vp]),
    "svit_cast_f32_bf16": (i32, [vp, vp, i64, vp]),
    "svit_table_interp": (i32, [vp, i32, i32, vp, vp, vp, vp]),
    "svit_table_interp_batched": (i32, [vp, i32, i32, vp]),
    "svit_transpose_cast_batched": (i32, [vp, vp, vp, i32, i32, vp]),
    "svit_transpose_bf16_batched": (i32, [vp, vp, vp, i32, i32, vp]),
    "svit_scale_cast": (i32, [vp, vp, vp, i32, i64, i32, i32, i32, i32, vp]),
    "svit_pad_cast_rows": (i32, [vp, vp, i32, i32, i32, vp]),
    "svit_reduce_defer": (i32, [i32, vp]),
    "svit_reduce_flush": (i32, [vp]),
    "svit_reduce_reset": (i32, [vp]),
    "svit_layernorm_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, vp]),
    "svit_layernorm_bwd": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i64, i32, vp,
                                 i64, vp]),
    "svit_im2col_patch": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "svit_im2col_patch_u8": (i32, [vp, i64, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "svit_fill_special_tokens": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "svit_special_token_grads": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "svit_pool_ln_fwd": (i32, [C.POINTER(PoolArgs), vp]),
    "svit_pool_ln_bwd": (i32, [C.POINTER(PoolLnBwdArgs), vp]),
    "svit_pool_ln_fwd_qkv": (i32, [C.POINTER(PoolArgs), vp]),
    "svit_pool_ln_fwd_qkv_sel": (i32, [C.POINTER(PoolArgs), C.POINTER(vp), vp]),
    "svit_pool_weight_sel": (i32, [vp, vp, vp, i32, vp]),
    "svit_pool_ln_bwd_qkv": (i32, [C.POINTER(PoolLnBwdArgs), vp]),
    "svit_pool_conv_bwd_qkv": (i32, [C.POINTER(PoolDgradArgs), C.POINTER(PoolWgradArgs), vp]),
    "svit_pool_conv_bwd_workspace": (i64, [C.POINTER(PoolDgradArgs)]),
    "svit_relpos_q_fwd": (i32, [C.POINTER(RelqArgs), vp]),
    "svit_relpos_q_bwd": (i32, [C.POINTER(RelqBwdArgs), vp]),
    "svit_relpos_scatter": (i32, [C.POINTER(RelqScatterArgs), vp]),
    "svit_relpos_gather": (i32, [C.POINTER(RelqGatherArgs), vp]),
    "svit_attn_fwd": (i32, [C.POINTER(AttnFwdArgs), vp]),
    "svit_attn_bwd": (i32, [C.POINTER(AttnBwdArgs), vp]),
    "svit_attn_bwd_parts": (i32, [C.POINTER(AttnBwdArgs)]),
    "svit_maxpool_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "svit_maxpool_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "svit_maxpool_bwd_bf16": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "svit_sumsq": (i32, [vp, i64, vp, vp, i64, vp]),
    "svit_adamw_step": (i32, [vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, f32, f32, i32, f32, vp]),
    "svit_head_fwd": (i32, [C.POINTER(HeadArgs), vp]),
    "svit_head_bwd": (i32, [C.POINTER(HeadBwdArgs), vp]),
    "svit_haog_loss": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "svit_haog_loss_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "svit_ce_loss": (i32, [vp, vp, i32, i32, vp, vp, vp]),
    "svit_step_draws": (i32, [vp, vp, i32, i32, vp, i32, C.c_float, vp, vp]),
    "svit_ensemble_update": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "svit_topk_correct": (i32, [vp, vp, i32, i32, vp, i32, vp, vp, vp]),
    # diagnostics block of the header: knobs for tools/ (never called by the product path)
    "svit_debug_set": (i32, [i32, i32]),
    "svit_debug_set_tn": (i32, [i32, i32]),
    "svit_debug_set_tn_tile": (i32, [i32]),
    "svit_debug_set_pool": (i32, [i32, i32]),
    "svit_attn_debug_set": (i32, [i32, i32]),
    "svit_debug_reset": (i32, []),
    "svit_debug_pool_bwd_path": (i32, []),
}
EXPORTS = tuple(sorted(_SIGS))


class SvitHipError(RuntimeError):
    pass


def load():
    """Load the library (once).  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SvitHipError(
                "libsvit_hip.so not found at %s -- run `python -m svit_amd.build` "
                "(the SViT HIP path has no PyTorch/CPU fallback)" % LIB_PATH)
        if os.environ.get("SVIT_HIP_LIB"):
            import warnings
            warnings.warn("SVIT_HIP_LIB is set: loading the SViT HIP library from %s instead of the "
                          "in-tree build" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def stream():
    """torch's current HIP stream on the current device (the capture stream inside a graph)."""
    return C.c_void_p(_raw_stream(_cur_device()))


def ptr(t):
    if t is None:
        return None
    return t.data_ptr()


def check(rc, what):
    if rc != 0:
        kind = {-2: "bad shape", -3: "bad alignment/stride", -4: "bad argument"}.get(
            rc, "hipError_t %d" % rc)
        raise SvitHipError("%s failed: %s" % (what, kind))


_trace = None  # list of (name, start_event, end_event, meta) while bench.py profiles a step


def start_trace():
    global _trace
    _trace = []


def stop_trace():
    global _trace
    t, _trace = _trace, None
    return t


def mark(label):
    """phase marker in the kernel trace (no-op unless bench.py is tracing)."""
    if _trace is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        _trace.append(("mark:" + label, e, e, None))


def call(name, *args, meta=None):
    if _trace is None:
        rc = getattr(load(), name)(*args, stream())
        check(rc, name)
        return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = getattr(load(), name)(*args, stream())
    e1.record()
    check(rc, name)
    _trace.append((name, e0, e1, meta))
