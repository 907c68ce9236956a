// Diagnostic: issue cost (cycles per wave-instruction) of the VALU ops the depthwise stencils can
// be built from, one wave per SIMD and two waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int OP>
__global__ void rate_kernel(float* out, unsigned long long* cyc, int iters) {
  float a[8];
  f32x2 p[4];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 4; ++i) p[i] = {a[2 * i], a[2 * i + 1]};
  const uint32_t x = 0x3f803f80u + threadIdx.x, w = 0x3f000000u;
  const f32x2 xw = {1.0001f, 0.9999f}, ww = {0.5f, 0.25f};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (OP == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          a[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, x), __builtin_bit_cast(bf16x2_t, w), a[i], false);
      } else if (OP == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);
      } else if (OP == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(xw), "v"(ww));
      } else if (OP == 3) {     // unpack a bf16 pair + two fma
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float lo = __uint_as_float(x << 16), hi = __uint_as_float(x & 0xffff0000u);
          a[2 * i] = __builtin_fmaf(lo, 0.5f, a[2 * i]);
          a[2 * i + 1] = __builtin_fmaf(hi, 0.25f, a[2 * i + 1]);
          asm volatile("" : "+v"(a[2 * i]), "+v"(a[2 * i + 1]));
        }
      }
      if (OP != 2) for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i];
  for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

extern "C" int valu_rate(int op, int threads, int blocks, int iters, float* out, unsigned long long* cyc) {
  switch (op) {
    case 0: hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters); break;
    case 1: hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters); break;
    case 2: hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters); break;
    case 3: hipLaunchKernelGGL(rate_kernel<3>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters); break;
  }
  return (int)hipDeviceSynchronize();
}
