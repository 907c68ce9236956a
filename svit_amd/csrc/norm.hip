// LayerNorm forward/backward over the fp32 residual stream -- HBM-bound:
// one wave per row, float4 accesses, fp32 statistics; backward accumulates dgamma/dbeta in
// registers over a grid-stride loop and flushes them with one atomic per column per block.
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "../../include/svit_hip.h"

namespace {
constexpr int LN_MAX_CHUNKS = 3;  // C <= 768: at most 3 float4 per lane

// NCH = float4 chunks per lane.  HALF (C <= 128): every 32-lane half of a wave owns its own row,
// so the narrow stages (C = 96) keep 48 of 64 lanes busy instead of 24.  Each (half-)wave keeps
// TWO rows in flight: both rows' loads are issued before the first reduction.
template <bool HALF>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int o = HALF ? 16 : 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NCH, bool HALF>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     bf16_t* __restrict__ y16,
                                                     float* __restrict__ y32,
                                                     float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int64_t rows,
                                                     int C, float eps) {
  constexpr int LW = HALF ? 32 : 64;          // lanes per row
  constexpr int RPW = HALF ? 2 : 1;           // rows per wave pass
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & (LW - 1), sub = HALF ? (lane >> 5) : 0;
  const int nch = C >> 2;
  const float inv_c = 1.0f / (float)C;
  float4 g[NCH], bt[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lr + i * LW;
    g[i] = bt[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nch) { g[i] = ((const float4*)gamma)[c]; bt[i] = ((const float4*)beta)[c]; }
  }
  const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
  for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RPW + sub; row0 < rows; row0 += 2 * stride) {
    float4 v[2][NCH];
    int64_t rk[2];
    bool has[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      has[k] = row0 + k * stride < rows;
      rk[k] = has[k] ? row0 + k * stride : row0;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lr + i * LW;
        v[k][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nch) v[k][i] = ((const float4*)(x + rk[k] * C))[c];
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) s += v[k][i].x + v[k][i].y + v[k][i].z + v[k][i].w;
      const float mean = row_sum<HALF>(s) * inv_c;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (lr + i * LW < nch) {
          const float a = v[k][i].x - mean, b = v[k][i].y - mean, cc = v[k][i].z - mean,
                      d = v[k][i].w - mean;
          sq += a * a + b * b + cc * cc + d * d;
        }
      }
      const float rstd = rsqrtf(row_sum<HALF>(sq) * inv_c + eps);
      if (!has[k]) continue;                  // uniform per (half-)wave; shuffles are done
      if (lr == 0) {
        if (mean_out) mean_out[rk[k]] = mean;
        if (rstd_out) rstd_out[rk[k]] = rstd;
      }
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lr + i * LW;
        if (c < nch) {
          float4 o;
          o.x = (v[k][i].x - mean) * rstd * g[i].x + bt[i].x;
          o.y = (v[k][i].y - mean) * rstd * g[i].y + bt[i].y;
          o.z = (v[k][i].z - mean) * rstd * g[i].z + bt[i].z;
          o.w = (v[k][i].w - mean) * rstd * g[i].w + bt[i].w;
          if (y16) {
            uint2 pk;
            pk.x = pack_bf16x2(o.x, o.y);
            pk.y = pack_bf16x2(o.z, o.w);
            *(uint2*)(y16 + rk[k] * C + c * 4) = pk;
          }
          if (y32) ((float4*)(y32 + rk[k] * C))[c] = o;
        }
      }
    }
  }
}

// Backward: same row mapping as the forward (NCH chunks per lane, HALF = one row per 32-lane
// half for C <= 128), two rows in flight per (half-)wave.
template <int NCH, bool HALF, bool DYB>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy_,
                                                     const float* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in,
                                                     const float* __restrict__ dres,
                                                     float* __restrict__ dx,
                                                     bf16_t* __restrict__ dx16,
                                                     const float* __restrict__ row_scale,
                                                     int rows_per_sample,
                                                     float* __restrict__ partial,
                                                     int64_t rows, int C) {
  constexpr int LW = HALF ? 32 : 64;
  constexpr int RPW = HALF ? 2 : 1;
  __shared__ float red[4][2][NCH * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & (LW - 1), sub = HALF ? (lane >> 5) : 0;
  const int nch = C >> 2;
  const float inv_c = 1.0f / (float)C;
  float4 gacc[NCH], bacc[NCH], gm[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    gacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    bacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = lr + i * LW;
    gm[i] = (c < nch) ? ((const float4*)gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
  for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RPW + sub; row0 < rows; row0 += 2 * stride) {
    float4 xv[2][NCH], dv[2][NCH], rv[2][NCH];
    float mean[2], rstd[2], sc16[2];
    int64_t rowk[2];
    bool has[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      has[k] = row0 + k * stride < rows;
      rowk[k] = has[k] ? row0 + k * stride : row0;
      mean[k] = mean_in[rowk[k]];
      rstd[k] = rstd_in[rowk[k]];
      sc16[k] = (dx16 && row_scale) ? row_scale[rowk[k] / rows_per_sample] : 1.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lr + i * LW;
        xv[k][i] = dv[k][i] = rv[k][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nch) {
          xv[k][i] = ((const float4*)(x + rowk[k] * C))[c];
          if (DYB) {      // the dgrad GEMM wrote bf16 (as autocast does): half the bytes
            const uint2 h = ((const uint2*)((const bf16_t*)dy_ + rowk[k] * C))[c];
            dv[k][i] = make_float4(lo_bf16(h.x), hi_bf16(h.x), lo_bf16(h.y), hi_bf16(h.y));
          } else {
            dv[k][i] = ((const float4*)((const float*)dy_ + rowk[k] * C))[c];
          }
          if (dres) rv[k][i] = ((const float4*)(dres + rowk[k] * C))[c];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float4 xh[NCH], g[NCH];
      float s1 = 0.f, s2 = 0.f;
      const float live = has[k] ? 1.f : 0.f;    // a clamped duplicate row must not be counted
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const float4 d = dv[k][i];
        xh[i].x = (xv[k][i].x - mean[k]) * rstd[k]; xh[i].y = (xv[k][i].y - mean[k]) * rstd[k];
        xh[i].z = (xv[k][i].z - mean[k]) * rstd[k]; xh[i].w = (xv[k][i].w - mean[k]) * rstd[k];
        if (lr + i * LW >= nch) xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        gacc[i].x += live * d.x * xh[i].x; gacc[i].y += live * d.y * xh[i].y;
        gacc[i].z += live * d.z * xh[i].z; gacc[i].w += live * d.w * xh[i].w;
        bacc[i].x += live * d.x; bacc[i].y += live * d.y;
        bacc[i].z += live * d.z; bacc[i].w += live * d.w;
        g[i].x = d.x * gm[i].x; g[i].y = d.y * gm[i].y;
        g[i].z = d.z * gm[i].z; g[i].w = d.w * gm[i].w;
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
      }
      s1 = row_sum<HALF>(s1) * inv_c;
      s2 = row_sum<HALF>(s2) * inv_c;
      if (!has[k]) continue;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lr + i * LW;
        if (c < nch) {
          float4 o;
          o.x = rstd[k] * (g[i].x - s1 - xh[i].x * s2) + rv[k][i].x;
          o.y = rstd[k] * (g[i].y - s1 - xh[i].y * s2) + rv[k][i].y;
          o.z = rstd[k] * (g[i].z - s1 - xh[i].z * s2) + rv[k][i].z;
          o.w = rstd[k] * (g[i].w - s1 - xh[i].w * s2) + rv[k][i].w;
          ((float4*)(dx + rowk[k] * C))[c] = o;
          if (dx16) {   // the next GEMM's operand: bf16(DropPath scale * dx), saves a cast pass
            uint2 h;
            h.x = pack_bf16x2(o.x * sc16[k], o.y * sc16[k]);
            h.y = pack_bf16x2(o.z * sc16[k], o.w * sc16[k]);
            ((uint2*)(dx16 + rowk[k] * C))[c] = h;
          }
        }
      }
    }
  }
  // the two halves of a wave hold partial sums of the same columns in HALF mode
  if (HALF) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      gacc[i].x += __shfl_xor(gacc[i].x, 32, 64); gacc[i].y += __shfl_xor(gacc[i].y, 32, 64);
      gacc[i].z += __shfl_xor(gacc[i].z, 32, 64); gacc[i].w += __shfl_xor(gacc[i].w, 32, 64);
      bacc[i].x += __shfl_xor(bacc[i].x, 32, 64); bacc[i].y += __shfl_xor(bacc[i].y, 32, 64);
      bacc[i].z += __shfl_xor(bacc[i].z, 32, 64); bacc[i].w += __shfl_xor(bacc[i].w, 32, 64);
    }
  }
  // block reduction of the per-wave dgamma/dbeta partials, then one partial row per block
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lr + i * LW;
    if (c < nch && sub == 0) {
      *(float4*)&red[wave][0][c * 4] = gacc[i];
      *(float4*)&red[wave][1][c * 4] = bacc[i];
    }
  }
  __syncthreads();
  float* prow = partial + (size_t)blockIdx.x * 2 * C;   // [dgamma | dbeta] row of this block
  for (int c = threadIdx.x; c < C; c += 256) {
    prow[c] = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
    prow[C + c] = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
  }
}
}  // namespace

extern "C" int svit_layernorm_fwd(const float* x, const float* gamma, const float* beta,
                                  void* y_bf16, float* y_f32, float* mean, float* rstd,
                                  int64_t rows, int C, float eps, void* stream) {
  if (!x || !gamma || !beta || (!y_bf16 && !y_f32)) return SVIT_ERR_ARG;
  if (rows <= 0 || C <= 0 || C % 4 != 0 || C > 768) return SVIT_ERR_SHAPE;
  constexpr int rpb_f = 8;
  int64_t blocks = (rows + rpb_f - 1) / rpb_f;            // two rows in flight per (half-)wave
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
#define SVIT_LN_FWD(NCH, HALF)                                                                \
  hipLaunchKernelGGL((ln_fwd_kernel<NCH, HALF>), dim3((unsigned)blocks), dim3(256), 0,           \
                     (hipStream_t)stream, x, gamma, beta, (bf16_t*)y_bf16, y_f32, mean, rstd,    \
                     rows, C, eps)
  if (C <= 128) SVIT_LN_FWD(1, true);
  else if (C <= 256) SVIT_LN_FWD(1, false);
  else if (C <= 512) SVIT_LN_FWD(2, false);
  else SVIT_LN_FWD(3, false);
#undef SVIT_LN_FWD
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_layernorm_bwd(const void* dy, int dy_is_bf16, const float* x, const float* gamma,
                                  const float* mean, const float* rstd, const float* dres,
                                  float* dx, void* dx_bf16, const float* row_scale,
                                  int rows_per_sample, float* dgamma, float* dbeta, int64_t rows,
                                  int C, float* workspace, int64_t workspace_floats,
                                  void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace)
    return SVIT_ERR_ARG;
  if (dx_bf16 && row_scale && rows_per_sample <= 0) return SVIT_ERR_ARG;
  if (rows <= 0 || C <= 0 || C % 4 != 0 || C > 768) return SVIT_ERR_SHAPE;
  // one partial row per block feeds the reduce launch: few rows for small inputs (the reduce
  // is latency-bound on its row count), up to 2048 blocks for the big ones (bandwidth)
  // rows per workgroup: 16 (round 4).  At 32 the 13064-row launches of the 14x14 stage were 409 workgroups = 6 waves per
  // CU, too few bytes in flight for the HBM rate; in the step 16 and 8 are level and 0.10 ms ahead of 32
  // (12.70 / 12.71 vs 12.81 ms; an isolated loop, whose operands sit in the Infinity Cache, shows no difference --
  // the round-3 sweep).
  constexpr int rpb = 16;
  int64_t blocks = (rows + rpb - 1) / rpb;
  if (blocks < 256) blocks = (rows + 3) / 4 < 256 ? (rows + 3) / 4 : 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks > workspace_floats / (2 * C)) blocks = workspace_floats / (2 * C);
  if (blocks < 1) return SVIT_ERR_ARG;
#define SVIT_LN_BWD(NCH, HALF)                                                                  \
  do {                                                                                            \
    if (dy_is_bf16)                                                                               \
      hipLaunchKernelGGL((ln_bwd_kernel<NCH, HALF, true>), dim3((unsigned)blocks), dim3(256), 0,  \
                         (hipStream_t)stream, dy, x, gamma, mean, rstd, dres, dx,                 \
                         (bf16_t*)dx_bf16, row_scale, rows_per_sample, workspace, rows, C);       \
    else                                                                                          \
      hipLaunchKernelGGL((ln_bwd_kernel<NCH, HALF, false>), dim3((unsigned)blocks), dim3(256), 0, \
                         (hipStream_t)stream, dy, x, gamma, mean, rstd, dres, dx,                 \
                         (bf16_t*)dx_bf16, row_scale, rows_per_sample, workspace, rows, C);       \
  } while (0)
  if (C <= 128) SVIT_LN_BWD(1, true);
  else if (C <= 256) SVIT_LN_BWD(1, false);
  else if (C <= 512) SVIT_LN_BWD(2, false);
  else SVIT_LN_BWD(3, false);
#undef SVIT_LN_BWD
  SVIT_LAUNCH_CHECK();
  SvitReduceDst dst = {{dgamma, dbeta, dbeta, dbeta, dbeta, dbeta}, {C, 2 * C, 2 * C, 2 * C, 2 * C, 2 * C}};
  svit_launch_reduce(workspace, (int)blocks, 2 * C, dst, (hipStream_t)stream);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}
