"""Wall-clock anatomy of one workgroup of the slab pooling forward: builds pool.hip (+misc.hip for the
reduce helper) with -DSVIT_POOL_STAMPS, runs one block shape, prints the phases in us."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "libpool_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
from svit_amd import build as B
pre = os.path.join(ROOT, "tools", "diag", "libsvit_diag_poolstamps.so")     # python tools/diag/build_variant.py poolstamps pool.hip -DSVIT_POOL_STAMPS
if os.path.exists(pre) and os.path.getmtime(pre) >= os.path.getmtime(os.path.join(B.CSRC, "pool.hip")):
    out = pre
else:
    srcs = [os.path.join(B.CSRC, f) for f in B.SOURCES]
    subprocess.check_call([B.HIPCC] + B.FLAGS + ["-DSVIT_POOL_STAMPS", "-shared"] + srcs + ["-o", out])
os.environ["SVIT_HIP_LIB"] = out
import runpy
sys.argv = ["pool_one.py"] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "tools", "pool_one.py"), run_name="__main__")
lib = ctypes.CDLL(out)
buf = (ctypes.c_ulonglong * 16)()
assert lib.svit_debug_pool_stamps(buf, 16) == 0
s = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
us = lambda i, j: (s[j] - s[i]) / 100.0
print("slab fwd WG (q, cg 0, chunk 0, bh 3): decode %.2f | fill issue %.2f | gain %.2f | dma wait %.2f | barrier %.2f | conv %.2f | special %.2f | total %.2f us"
      % (us(0, 1) * 0, us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(0, 6)))
if s[8] and s[13]:
    print("mfma fwd WG (q, cb 0, bh 3): zero halos + fill %.2f | gains + Toeplitz fragments %.2f | barrier %.2f | units %.2f | cls / objects %.2f | total %.2f us"
          % (us(8, 9), us(9, 10), us(10, 11), us(11, 12), us(12, 13), us(8, 13)))
if hasattr(lib, "svit_debug_pool_wg_times"):
    n = 4 * 2048
    wb = (ctypes.c_ulonglong * n)()
    assert lib.svit_debug_pool_wg_times(wb, n) == 0
    w = np.frombuffer(wb, dtype=np.uint64).astype(np.int64).reshape(2048, 4)
    w = w[w[:, 1] > 0]
    t0 = w[:, 0].min()
    st, en = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0
    print("mfma fwd kernel: %d workgroups; starts %.1f .. %.1f us (median %.1f), ends %.1f .. %.1f us; duration q-kind median %.1f us (max %.1f), kv-kind median %.1f us (max %.1f)"
          % (len(w), st.min(), st.max(), np.median(st), en.min(), en.max(),
             np.median((en - st)[w[:, 2] == 0]), (en - st)[w[:, 2] == 0].max(), np.median((en - st)[w[:, 2] == 1]), (en - st)[w[:, 2] == 1].max()))
    late = st > 0.5 * en.max()
    print("   workgroups that START in the second half of the kernel: %d (of %d)" % (late.sum(), len(w)))
    hw = w[:, 3]
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    print("   distinct (se, sh, cu) triples seen: %d" % len(set(zip(se.tolist(), sh.tolist(), cu.tolist()))))
