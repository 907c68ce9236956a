#!/usr/bin/env python3
"""Builds tools/diag/slp_repro.hip and runs its four packed-multiply forms alone and beside the
library's TN GEMM (second stream), several rounds; prints the number of wrong results per form.
    python tools/diag/run_slp_repro.py   (on the GPU box)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from svit_amd import ops

out = os.path.join(ROOT, "gpurun_out", "libslp_repro.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-fno-slp-vectorize", "-shared",
                       os.path.join(ROOT, "tools", "diag", "slp_repro.hip"), "-o", out])
lib = C.CDLL(out)
lib.slp_repro_run.restype = C.c_int
lib.slp_repro_run.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
bad = torch.zeros(8, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
M, N, K = 13064, 1152, 384
a, b = torch.randn(M, N, device="cuda").bfloat16(), torch.randn(M, K, device="cuda").bfloat16()
dw = torch.zeros(N, K, device="cuda")
names = ["in place op_sel:[0,1]", "in place op_sel:[0,1] op_sel_hi:[1,0]", "dest!=src op_sel:[0,1]", "scalar v_mul x2",
         "sgpr op_sel_hi:[1,0]", "plain v_pk_mul"]
from svit_amd import hip
w = torch.randn(N, K, device="cuda").bfloat16()
a_nt = torch.randn(M, K, device="cuda").bfloat16()
bias = torch.zeros(N, device="cuda")
qa, ka, v = (torch.randn(8, 4, 1633, 128, device="cuda").bfloat16(), (torch.randn(8, 4, 457, 128, device="cuda") * 0.15).bfloat16(),
             torch.randn(8, 4, 457, 96, device="cuda").bfloat16())
x32 = torch.randn(13064, 384, device="cuda")
g32, b32 = torch.ones(384, device="cuda"), torch.zeros(384, device="cuda")
partners = {
    "alone": None,
    "TN GEMM": lambda: ops.gemm_tn(a, b, dw),
    "NT GEMM": lambda: ops.gemm_nt(a_nt, w, bias, hip.EPI_BF16),
    "attention fwd": lambda: ops.attn_fwd(qa, ka, v, 96 ** -0.5),
    "LayerNorm fwd": lambda: ops.layernorm_fwd(x32, g32, b32),
    "torch matmul (hipBLASLt)": lambda: torch.mm(a_nt, w.t()),
    "torch elementwise": lambda: x32.mul(1.0001),
}
for pname, fn in partners.items():
    bad.zero_()
    torch.cuda.synchronize()
    for rnd in range(10):
        if fn is not None:
            with torch.cuda.stream(side):
                for _ in range(6):
                    fn()
        for form in range(6):
            rc = lib.slp_repro_run(form, bad.data_ptr(), 1024, 4000, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
    torch.cuda.synchronize()
    print("%-26s wrong results per form (of %.1e each):" % (pname, 10 * 1024 * 256 * 4000 * 2),
          {names[i]: int(bad[i]) for i in range(6)}, flush=True)
