// Standalone reproducer for the round-2 finding "in-place cross-half v_pk_mul_f32 returns wrong
// halves beside TN-GEMM waves" (profiles/r02_wgrad_overlap_rootcause.md).  One kernel, four forms of
// the packed multiply written in inline asm so that the encoding is exactly the one under test:
//   form 0: v_pk_mul_f32 d, a, d op_sel:[0,1]                 in place, low result reads src1.hi
//   form 1: v_pk_mul_f32 d, a, d op_sel:[0,1] op_sel_hi:[1,0] in place, both results cross
//   form 2: v_pk_mul_f32 e, a, d op_sel:[0,1]                 same reads, destination != source
//   form 3: two scalar v_mul_f32                              reference instruction mix
//   form 4: v_pk_mul_f32 e, a, s[..] op_sel_hi:[1,0]          scalar broadcast (what the shipped library contains)
//   form 5: v_pk_mul_f32 e, a, d                              no cross-half selection at all
// Every lane repeats the operation `iters` times on values it can predict exactly (powers of two)
// and counts results that differ from the prediction; the host runs the kernel alone and beside
// the library's TN GEMM on a second stream (tools/diag/run_slp_repro.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int FORM>
__global__ __launch_bounds__(256) void pk_kernel(unsigned* __restrict__ bad, int iters) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x;
  unsigned errs = 0;
  for (int it = 0; it < iters; ++it) {
    // a = (2, 4), d = (8, 16) scaled by a lane / iteration dependent power of two: products exact
    // (every input is built with scalar instructions only, so only the operation under test is packed)
    const float sc = __builtin_ldexpf(1.0f, (lane + it) & 7);
    float ax = 2.0f * sc, ay = 4.0f * sc;
    asm volatile("" : "+v"(ax), "+v"(ay));
    f2 a = {ax, ay}, d = {8.0f, 16.0f}, e = {0.f, 0.f};
    asm volatile("" : "+v"(a), "+v"(d));
    float lo, hi, want_lo, want_hi;
    if (FORM == 0) {
      asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1]" : "+v"(d) : "v"(a));
      lo = d.x; hi = d.y; want_lo = a.x * 16.0f; want_hi = a.y * 16.0f;
    } else if (FORM == 1) {
      asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(d) : "v"(a));
      lo = d.x; hi = d.y; want_lo = a.x * 16.0f; want_hi = a.y * 8.0f;
    } else if (FORM == 2) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(e) : "v"(a), "v"(d));
      lo = e.x; hi = e.y; want_lo = a.x * 16.0f; want_hi = a.y * 16.0f;
    } else if (FORM == 4) {
      asm volatile("s_mov_b32 s20, 0x41800000\n\ts_mov_b32 s21, 0x41000000\n\t"
                   "v_pk_mul_f32 %0, %1, s[20:21] op_sel_hi:[1,0]" : "=&v"(e) : "v"(a) : "s20", "s21");
      lo = e.x; hi = e.y; want_lo = a.x * 16.0f; want_hi = a.y * 16.0f;
    } else if (FORM == 5) {
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(e) : "v"(a), "v"(d));
      lo = e.x; hi = e.y; want_lo = a.x * 8.0f; want_hi = a.y * 16.0f;
    } else {
      lo = a.x * d.y; hi = a.y * d.y; want_lo = a.x * 16.0f; want_hi = a.y * 16.0f;
      asm volatile("" : "+v"(lo), "+v"(hi));
    }
    errs += (lo != want_lo) + (hi != want_hi);
  }
  if (errs) atomicAdd(bad + FORM, errs);
}

extern "C" int slp_repro_run(int form, unsigned* bad, int blocks, int iters, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (form) {
    case 0: hipLaunchKernelGGL(pk_kernel<0>, dim3(blocks), dim3(256), 0, st, bad, iters); break;
    case 1: hipLaunchKernelGGL(pk_kernel<1>, dim3(blocks), dim3(256), 0, st, bad, iters); break;
    case 2: hipLaunchKernelGGL(pk_kernel<2>, dim3(blocks), dim3(256), 0, st, bad, iters); break;
    case 4: hipLaunchKernelGGL(pk_kernel<4>, dim3(blocks), dim3(256), 0, st, bad, iters); break;
    case 5: hipLaunchKernelGGL(pk_kernel<5>, dim3(blocks), dim3(256), 0, st, bad, iters); break;
    default: hipLaunchKernelGGL(pk_kernel<3>, dim3(blocks), dim3(256), 0, st, bad, iters); break;
  }
  return (int)hipGetLastError();
}
