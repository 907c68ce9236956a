// Operand layout probe for v_mfma_f32_4x4x4_16b_bf16 (16 independent 4x4x4 products per wave) on gfx950.
// Candidate (CDNA3 ISA, 4x4x4 forms): block = lane / 4; A[i][k]: lane 4 block + i, element k; B[k][j]: lane 4 block + j,
// element k; D[i][j]: lane 4 block + j, register i.  The host checks it with exact small integers (asymmetric in i, j, k
// and the block).  hipcc --offload-arch=gfx950 -O3 -shared -fPIC mfma4x4_layout.hip -o libmfma4x4_layout.so
#include <hip/hip_runtime.h>
#include <cstdint>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe_kernel(const uint16_t* a, const uint16_t* b, float* d) {   // a, b: [64 lanes][4 elements]
  const int l = threadIdx.x;
  s4 av, bv;
  for (int e = 0; e < 4; ++e) { av[e] = (short)a[l * 4 + e]; bv[e] = (short)b[l * 4 + e]; }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av, bv, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}
extern "C" int mfma4x4_probe(const void* a, const void* b, void* d, void* stream) {
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)a, (const uint16_t*)b, (float*)d);
  return (int)hipGetLastError();
}
