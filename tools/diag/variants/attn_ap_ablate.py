#!/usr/bin/env python3
"""Anatomy of the anti-phase attention forward by ablation (diagnostic variants libsvit_diag_apabl<mask>.so, built with
-DSVIT_ATTN_STAMPS=0 -DSVIT_AP_ABL=<mask>): loop cycles per tile of workgroup 0 for every variant given.
    python tools/diag/attn_ap_ablate.py Nq Nk DA heads tag [tag ...]        (GPU box; one subprocess per variant)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 6 or (len(sys.argv) == 6 and os.environ.get("AP_CHILD") != "1"):
    for tag in sys.argv[5:]:
        env = dict(os.environ, AP_CHILD="1")
        subprocess.run([sys.executable, __file__] + sys.argv[1:5] + [tag], env=env)
    sys.exit(0)
Nq, Nk, DA, h = [int(v) for v in sys.argv[1:5]]
tag = sys.argv[5]
lib_path = os.path.join(ROOT, "tools", "diag", "libsvit_diag_%s.so" % tag)
os.environ["SVIT_HIP_LIB"] = lib_path
import warnings
warnings.simplefilter("ignore")
import numpy as np, torch
from svit_amd import hip, ops
lib = hip.load()
B, J = 8, (22 if DA == 128 else 36)
qa = (torch.randn(B, h, Nq, DA, device="cuda") * 0.5).bfloat16()
ka = (torch.randn(B, h, Nk, DA, device="cuda") * 0.5 * 0.1472).bfloat16()
v = torch.randn(B, h, Nk, 96, device="cuda").bfloat16()
qa[..., 96 + J:] = 0
ka[..., 96 + J:] = 0
assert lib.svit_attn_debug_set(4, 1) == 0 and lib.svit_attn_debug_set(5, 0) == 0
for _ in range(3):
    ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=J)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attn_fwd(qa, ka, v, 96 ** -0.5, bias_cols=J)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
raw = ctypes.CDLL(lib_path)
n = 2 * 64 * 8 + 8
buf = (ctypes.c_ulonglong * n)()
assert raw.svit_debug_attn_ap_stamps(buf, n) == 0
s = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
nt = (Nk + 63) // 64
print("%-28s launch %6.1f us | workgroup 0: entry -> loop end %7d cycles (%.2f GHz) = %5.0f per tile; stores +%d"
      % (tag, us, s[1026] - s[1024], (s[1026] - s[1024]) / ((s[1027] - s[1025]) / 100e6) / 1e9, (s[1026] - s[1024]) / nt,
         s[1028] - s[1026]), flush=True)
