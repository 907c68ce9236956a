// LayerNorm forward/backward over the fp32 residual stream -- HBM-bound:
// one wave per row, float4 accesses, fp32 statistics; backward accumulates dgamma/dbeta in
// registers over a grid-stride loop and flushes them with one atomic per column per block.
#include "common.h"
#include "../../include/svit_hip.h"

namespace {
constexpr int LN_MAX_CHUNKS = 3;  // C <= 768: at most 3 float4 per lane

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     bf16_t* __restrict__ y16,
                                                     float* __restrict__ y32,
                                                     float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int64_t rows,
                                                     int C, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = C >> 2;
  const float inv_c = 1.0f / (float)C;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float4* xr = (const float4*)(x + row * C);
    float4 v[LN_MAX_CHUNKS];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
      const int c = lane + i * 64;
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < nch) v[i] = xr[c];
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = wave_sum(s) * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        sq += a * a + b * b + cc * cc + d * d;
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) * inv_c + eps);
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
        const float4 g = ((const float4*)gamma)[c], b = ((const float4*)beta)[c];
        float4 o;
        o.x = (v[i].x - mean) * rstd * g.x + b.x;
        o.y = (v[i].y - mean) * rstd * g.y + b.y;
        o.z = (v[i].z - mean) * rstd * g.z + b.z;
        o.w = (v[i].w - mean) * rstd * g.w + b.w;
        if (y16) {
          uint2 pk;
          pk.x = pack_bf16x2(o.x, o.y);
          pk.y = pack_bf16x2(o.z, o.w);
          *(uint2*)(y16 + row * C + c * 4) = pk;
        }
        if (y32) ((float4*)(y32 + row * C))[c] = o;
      }
    }
  }
}

__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy,
                                                     const float* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in,
                                                     const float* __restrict__ dres,
                                                     float* __restrict__ dx,
                                                     float* __restrict__ partial,
                                                     int64_t rows, int C) {
  __shared__ float red[4][2][768];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = C >> 2;
  const float inv_c = 1.0f / (float)C;
  float4 gacc[LN_MAX_CHUNKS], bacc[LN_MAX_CHUNKS], gm[LN_MAX_CHUNKS];
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    gacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    bacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = lane + i * 64;
    gm[i] = (c < nch) ? ((const float4*)gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    float4 xh[LN_MAX_CHUNKS], g[LN_MAX_CHUNKS];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
      const int c = lane + i * 64;
      xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < nch) {
        const float4 xv = ((const float4*)(x + row * C))[c];
        const float4 d = ((const float4*)(dy + row * C))[c];
        xh[i].x = (xv.x - mean) * rstd; xh[i].y = (xv.y - mean) * rstd;
        xh[i].z = (xv.z - mean) * rstd; xh[i].w = (xv.w - mean) * rstd;
        gacc[i].x += d.x * xh[i].x; gacc[i].y += d.y * xh[i].y;
        gacc[i].z += d.z * xh[i].z; gacc[i].w += d.w * xh[i].w;
        bacc[i].x += d.x; bacc[i].y += d.y; bacc[i].z += d.z; bacc[i].w += d.w;
        g[i].x = d.x * gm[i].x; g[i].y = d.y * gm[i].y;
        g[i].z = d.z * gm[i].z; g[i].w = d.w * gm[i].w;
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
      }
    }
    s1 = wave_sum(s1) * inv_c;
    s2 = wave_sum(s2) * inv_c;
#pragma unroll
    for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
      const int c = lane + i * 64;
      if (c < nch) {
        float4 o;
        o.x = rstd * (g[i].x - s1 - xh[i].x * s2);
        o.y = rstd * (g[i].y - s1 - xh[i].y * s2);
        o.z = rstd * (g[i].z - s1 - xh[i].z * s2);
        o.w = rstd * (g[i].w - s1 - xh[i].w * s2);
        if (dres) {
          const float4 r = ((const float4*)(dres + row * C))[c];
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        ((float4*)(dx + row * C))[c] = o;
      }
    }
  }
  // block reduction of the per-wave dgamma/dbeta partials, then one atomic per column
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
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
  int64_t blocks = (rows + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                     gamma, beta, (bf16_t*)y_bf16, y_f32, mean, rstd, rows, C, eps);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_layernorm_bwd(const float* dy, const float* x, const float* gamma,
                                  const float* mean, const float* rstd, const float* dres,
                                  float* dx, float* dgamma, float* dbeta, int64_t rows, int C,
                                  float* workspace, int64_t workspace_floats, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace)
    return SVIT_ERR_ARG;
  if (rows <= 0 || C <= 0 || C % 4 != 0 || C > 768) return SVIT_ERR_SHAPE;
  // one partial row per block feeds the reduce launch: few rows for small inputs (the reduce
  // is latency-bound on its row count), up to 2048 blocks for the big ones (bandwidth)
  int64_t blocks = (rows + 31) / 32;
  if (blocks < 256) blocks = (rows + 3) / 4 < 256 ? (rows + 3) / 4 : 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks > workspace_floats / (2 * C)) blocks = workspace_floats / (2 * C);
  if (blocks < 1) return SVIT_ERR_ARG;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy,
                     x, gamma, mean, rstd, dres, dx, workspace, rows, C);
  SVIT_LAUNCH_CHECK();
  SvitReduceDst dst = {{dgamma, dbeta, dbeta, dbeta, dbeta, dbeta}, {C, 2 * C, 2 * C, 2 * C, 2 * C, 2 * C}};
  svit_launch_reduce(workspace, (int)blocks, 2 * C, dst, (hipStream_t)stream);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}
