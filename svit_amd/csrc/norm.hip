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

// NCH = float4 chunks per lane (C <= 256 * NCH).  Each wave keeps TWO rows in flight: the loads
// of both are issued before the first reduction, which halves the number of exposed memory
// round trips of this latency-bound kernel (rows / (4 * gridDim.x) iterations per wave).
template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy,
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
  __shared__ float red[4][2][NCH * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = C >> 2;
  const float inv_c = 1.0f / (float)C;
  float4 gacc[NCH], bacc[NCH], gm[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    gacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    bacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = lane + i * 64;
    gm[i] = (c < nch) ? ((const float4*)gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t row0 = (int64_t)blockIdx.x * 4 + wave; row0 < rows; row0 += 2 * stride) {
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
        const int c = lane + i * 64;
        xv[k][i] = dv[k][i] = rv[k][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nch) {
          xv[k][i] = ((const float4*)(x + rowk[k] * C))[c];
          dv[k][i] = ((const float4*)(dy + rowk[k] * C))[c];
          if (dres) rv[k][i] = ((const float4*)(dres + rowk[k] * C))[c];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (!has[k]) continue;                       // wave-uniform
      float4 xh[NCH], g[NCH];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const float4 d = dv[k][i];
        xh[i].x = (xv[k][i].x - mean[k]) * rstd[k]; xh[i].y = (xv[k][i].y - mean[k]) * rstd[k];
        xh[i].z = (xv[k][i].z - mean[k]) * rstd[k]; xh[i].w = (xv[k][i].w - mean[k]) * rstd[k];
        if (lane + i * 64 >= nch) xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        gacc[i].x += d.x * xh[i].x; gacc[i].y += d.y * xh[i].y;
        gacc[i].z += d.z * xh[i].z; gacc[i].w += d.w * xh[i].w;
        bacc[i].x += d.x; bacc[i].y += d.y; bacc[i].z += d.z; bacc[i].w += d.w;
        g[i].x = d.x * gm[i].x; g[i].y = d.y * gm[i].y;
        g[i].z = d.z * gm[i].z; g[i].w = d.w * gm[i].w;
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
      }
      s1 = wave_sum(s1) * inv_c;
      s2 = wave_sum(s2) * inv_c;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
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
  // block reduction of the per-wave dgamma/dbeta partials, then one partial row per block
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
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
  int64_t blocks = (rows + 31) / 32;
  if (blocks < 256) blocks = (rows + 3) / 4 < 256 ? (rows + 3) / 4 : 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks > workspace_floats / (2 * C)) blocks = workspace_floats / (2 * C);
  if (blocks < 1) return SVIT_ERR_ARG;
#define SVIT_LN_BWD(NCH)                                                                        \
  hipLaunchKernelGGL(ln_bwd_kernel<NCH>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, \
                     dy, x, gamma, mean, rstd, dres, dx, (bf16_t*)dx_bf16, row_scale,               \
                     rows_per_sample, workspace, rows, C)
  if (C <= 256) SVIT_LN_BWD(1);
  else if (C <= 512) SVIT_LN_BWD(2);
  else SVIT_LN_BWD(3);
#undef SVIT_LN_BWD
  SVIT_LAUNCH_CHECK();
  SvitReduceDst dst = {{dgamma, dbeta, dbeta, dbeta, dbeta, dbeta}, {C, 2 * C, 2 * C, 2 * C, 2 * C, 2 * C}};
  svit_launch_reduce(workspace, (int)blocks, 2 * C, dst, (hipStream_t)stream);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}
