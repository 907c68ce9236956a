"""Wrappers of the STREAMING conv backward of rounds 1-4 (svit_pool_conv_dgrad / _wgrad and their q-k-v forms).  Round 6: those kernels
exist only in a diagnostic build of csrc/pool.hip --
    python tools/diag/build_variant.py poolstream pool.hip -DSVIT_DIAG_POOL_STREAMING
    SVIT_HIP_LIB=tools/diag/libsvit_diag_poolstream.so python tools/...
-- the product library does not export them (include/svit_hip.h, diagnostics block).  Importing this module against the product
library exits with that message instead of an AttributeError deep inside a timing loop."""
import ctypes as C
import sys

from svit_amd import hip
from svit_amd.ops import _pool_dgrad_args, _pool_wgrad_args

if not hasattr(hip.load(), "svit_pool_conv_dgrad_qkv"):
    sys.exit("%s needs the streaming conv-backward kernels: build tools/diag/libsvit_diag_poolstream.so with "
             "`python tools/diag/build_variant.py poolstream pool.hip -DSVIT_DIAG_POOL_STREAMING` and set SVIT_HIP_LIB to it" % sys.argv[0])


def pool_conv_dgrad(dpre, conv_w, dqkv, which, B, heads, thw, n_obj, stride_hw):
    a = hip.PoolDgradArgs()
    _pool_dgrad_args(a, dpre, conv_w, dqkv, which, B, heads, thw, n_obj, stride_hw)
    hip.call("svit_pool_conv_dgrad", C.byref(a))


def pool_conv_dgrad_qkv(dpres, conv_ws, dqkv, B, heads, thw, n_obj, strides):
    arr = (hip.PoolDgradArgs * 3)()
    for i in range(3):
        _pool_dgrad_args(arr[i], dpres[i], conv_ws[i], dqkv, i, B, heads, thw, n_obj, strides[i])
    hip.call("svit_pool_conv_dgrad_qkv", arr)


def pool_conv_wgrad(dpre, qkv, which, dw, B, heads, thw, n_obj, stride_hw):
    a = hip.PoolWgradArgs()
    _pool_wgrad_args(a, dpre, qkv, which, dw, B, heads, thw, n_obj, stride_hw)
    hip.call("svit_pool_conv_wgrad", C.byref(a))


def pool_conv_wgrad_qkv(dpres, qkv, dws, B, heads, thw, n_obj, strides, ws=None):
    arr = (hip.PoolWgradArgs * 3)()
    for i in range(3):
        _pool_wgrad_args(arr[i], dpres[i], qkv, i, dws[i], B, heads, thw, n_obj, strides[i], ws)
    hip.call("svit_pool_conv_wgrad_qkv", arr)
