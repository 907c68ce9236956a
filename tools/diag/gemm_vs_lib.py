"""Diagnostic: svit_gemm_nt (plain bf16 epilogue) beside torch's library GEMM (hipBLASLt) on the
shapes that carry the step.  Not part of the product path; the library is only a yardstick."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from svit_amd import ops, hip

SHAPES = [(13064, 1536, 384), (13064, 384, 1536), (13064, 384, 1152), (13064, 1152, 384),
          (13064, 384, 384), (52256, 96, 96), (50696, 768, 192), (50696, 192, 768),
          (201224, 384, 96), (201224, 96, 384), (3656, 3072, 768), (3656, 768, 3072)]


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
  for M, N, K in SHAPES:
      a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
      w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
      b = torch.zeros(N, device="cuda", dtype=torch.float32)
      out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
      t_mine = timeit(lambda: ops.gemm_nt(a, w, b, hip.EPI_BF16, out=out))
      t_lib = timeit(lambda: torch.mm(a, w.t(), out=out))
      fl = 2.0 * M * N * K
      print(f"M {M:6d} N {N:5d} K {K:5d}  svit {t_mine:7.1f} us {fl/t_mine*1e-6:6.0f} TF   "
            f"library {t_lib:7.1f} us {fl/t_lib*1e-6:6.0f} TF", flush=True)


if __name__ == "__main__":
  main()
