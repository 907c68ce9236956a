// Diagnostic only (not part of libsvit_hip.so): does a workgroup's LDS stay intact while
// workgroups of ANOTHER kernel are co-resident on the same CU?  Each workgroup fills `bytes` of
// dynamic LDS with a pattern, then re-reads it `iters` times and reports the first mismatch.
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t pat(uint32_t block, uint32_t i) {
  return (block * 2654435761u) ^ (i * 40503u + 0x9e3779b9u);
}

extern "C" __global__ void lds_canary_kernel(int words, int iters, uint32_t* __restrict__ out) {
  extern __shared__ uint32_t lds[];
  const uint32_t blk = blockIdx.x;
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = pat(blk, i);
  __syncthreads();
  uint32_t bad = 0, first_off = 0xffffffffu, first_val = 0, first_it = 0;
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < words; i += blockDim.x) {
      const uint32_t v = lds[i];
      if (v != pat(blk, i)) {
        if (!bad) { first_off = i; first_val = v; first_it = it; }
        ++bad;
        lds[i] = pat(blk, i);      // repair, so that every corruption event is counted once
      }
    }
    __syncthreads();
  }
  // per-thread records: [bad, first_off, first_val, first_it]
  uint32_t* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  o[0] = bad; o[1] = first_off; o[2] = first_val; o[3] = first_it;
}

extern "C" int lds_canary_launch(int blocks, int threads, int bytes, int iters, uint32_t* out, void* stream) {
  hipFuncSetAttribute((const void*)lds_canary_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  hipLaunchKernelGGL(lds_canary_kernel, dim3(blocks), dim3(threads), bytes, (hipStream_t)stream,
                     bytes / 4, iters, out);
  return (int)hipGetLastError();
}
