// Shared epilogue of the NT GEMM kernels (gemm.hip, gemm_v2.hip) -- gfx950.
#pragma once
#include "common.h"
#include "../../include/svit_hip.h"

// Each wave transposes its accumulators, 16 rows x (32*NB) columns at a time, through a
// private LDS region so that global traffic is row-contiguous and vectorised (16 B fp32 / 8 B
// bf16 per lane) instead of 2-byte column-strided accesses.  Needs
// n_waves * 16 * (32*NB + 4) * 4 bytes of LDS at `smem`; all waves of the block must call it.
template <int RB, int NB, int EPI>
__device__ __forceinline__ void nt_epilogue(const svit_gemm_args& p, f32x16_t (&acc)[RB][NB],
                                            unsigned char* smem, int m0, int n0, int wm, int wn,
                                            int lane, int wave) {
  constexpr int WN = 32 * NB;
  constexpr int EP_LD = WN + 4;
  float* stg = (float*)smem + wave * (16 * EP_LD);
#pragma unroll
  for (int ih = 0; ih < 2 * RB; ++ih) {
    const int i = ih >> 1, half = ih & 1;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int rr = 0; rr < 8; ++rr)
        stg[((rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5)) * EP_LD + j * 32 + (lane & 31)] =
            acc[i][j][half * 8 + rr];
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < 2 * NB; ++it) {
      const int idx = lane + 64 * it;
      const int rl = idx / (8 * NB), c4 = idx % (8 * NB);
      const int row = m0 + wm * 32 * RB + i * 32 + half * 16 + rl;
      const int col = n0 + wn * WN + c4 * 4;
      if (row >= p.M || col >= p.N) continue;
      float4 v = *(const float4*)(stg + rl * EP_LD + c4 * 4);
      if (p.bias) {
        const float4 b = *(const float4*)(p.bias + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      if constexpr (EPI == SVIT_EPI_BF16) {
        uint2 o;
        o.x = pack_bf16x2(v.x, v.y); o.y = pack_bf16x2(v.z, v.w);
        *(uint2*)((bf16_t*)p.out + (size_t)row * p.ldo + col) = o;
      } else if constexpr (EPI == SVIT_EPI_GELU) {
        uint2 o;
        o.x = pack_bf16x2(v.x, v.y); o.y = pack_bf16x2(v.z, v.w);
        *(uint2*)((bf16_t*)p.out2 + (size_t)row * p.ldo2 + col) = o;
        o.x = pack_bf16x2(gelu_erf(v.x), gelu_erf(v.y));
        o.y = pack_bf16x2(gelu_erf(v.z), gelu_erf(v.w));
        *(uint2*)((bf16_t*)p.out + (size_t)row * p.ldo + col) = o;
      } else if constexpr (EPI == SVIT_EPI_RESID) {
        const float s = p.row_scale ? p.row_scale[row / p.rows_per_sample] : 1.f;
        const float4 res = *(const float4*)((const float*)p.aux + (size_t)row * p.ldaux + col);
        float4 o;
        o.x = res.x + s * v.x; o.y = res.y + s * v.y; o.z = res.z + s * v.z; o.w = res.w + s * v.w;
        *(float4*)((float*)p.out + (size_t)row * p.ldo + col) = o;
      } else if constexpr (EPI == SVIT_EPI_F32) {
        size_t orow = row;
        if (p.remap_L > 0)
          orow = (size_t)(row / p.remap_L) * p.remap_N + p.remap_off + (row % p.remap_L);
        float4* o = (float4*)((float*)p.out + orow * p.ldo + col);
        if (p.accumulate) {
          const float4 old = *o;
          v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
        }
        *o = v;
      } else if constexpr (EPI == SVIT_EPI_DGELU) {
        const uint2 h = *(const uint2*)((const bf16_t*)p.aux + (size_t)row * p.ldaux + col);
        uint2 o;
        o.x = pack_bf16x2(v.x * gelu_erf_grad(lo_bf16(h.x)), v.y * gelu_erf_grad(hi_bf16(h.x)));
        o.y = pack_bf16x2(v.z * gelu_erf_grad(lo_bf16(h.y)), v.w * gelu_erf_grad(hi_bf16(h.y)));
        *(uint2*)((bf16_t*)p.out + (size_t)row * p.ldo + col) = o;
      }
    }
    if (ih + 1 < 2 * RB) __syncthreads();
  }
}
