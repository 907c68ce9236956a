"""Image-rank step parity (tests/smoke_impl.compare_step, T = 1) with the one-plane pooling kernel on / off, alternating."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import smoke_impl as S
from svit_amd import hip
lib = hip.load()
for on in (1, 0, 1, 0):
    lib.svit_debug_set_pool(3, on)
    res = S.compare_step(4, 64, 3, image=True)
    print("frame kernel", on, "worst", res["grad_cos_worst"], res["grad_cos_worst_name"], "global", res["grad_cos_global"], flush=True)
lib.svit_debug_reset()
