// bf16 MFMA weight-gradient GEMM for the nn.Linear family -- gfx950 only.
//   svit_gemm_tn : dW[N,K] += A[M,N]^T * B[M,K]  (reduction over rows, operands consumed
//                  through ds_read_b64_tr_b16 transposed LDS reads; fused bias gradient)
// (the forward / dgrad kernel svit_gemm_nt lives in gemm_nt.hip)
#include "common.h"
#include "../../include/svit_hip.h"

namespace {

// ---------------------------------------------------------------------------------------
// TN kernel.  Every wave owns a 64(n) x 96(k) output tile (6 MFMAs per 10 transposed LDS
// reads); block = 4 waves as WAVES_N x WAVES_K:
//     2 x 2 -> 128(n) x 192(k) tile   (K >= 192)
//     4 x 1 -> 256(n) x  96(k) tile   (K = 96: blocks 0-1, rel-pos tables, ...)
// 32 reduction rows per step, register-staged double buffer, one barrier per step.  Both
// operands sit [m][cols] in LDS and are read transposed (ds_read_b64_tr_b16); row strides are
// chosen so that the 4 rows of a transposed block fall on disjoint 64-byte bank groups.
// ---------------------------------------------------------------------------------------
constexpr int TN_BM = 32;        // reduction rows per step

template <int WAVES_N, int WAVES_K>
struct TnCfg {
  static constexpr int TN = 64 * WAVES_N, TK = 96 * WAVES_K;
  // row strides: (bytes mod 256) must be 64 or 192 so rows r..r+3 start in 4 different groups
  static constexpr int ROWA = TN * 2 + 64;                       // 320 / 576
  static constexpr int ROWB = (TK * 2) % 256 == 128 ? TK * 2 + 64 : TK * 2;   // 192 / 448
  static constexpr int STAGE = TN_BM * (ROWA + ROWB);
};

template <int WAVES_N, int WAVES_K>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A, int lda,
                                                         const bf16_t* __restrict__ B, int ldb,
                                                         float* __restrict__ dW, int lddw, int M,
                                                         int N, int K, int rows_per_split,
                                                         float* __restrict__ dbias) {
  using C = TnCfg<WAVES_N, WAVES_K>;
  constexpr int TN = C::TN, TK = C::TK, ROWA = C::ROWA, ROWB = C::ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][C::STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WAVES_K, wk = wave % WAVES_K;
  const int n0 = blockIdx.x * TN, k0 = blockIdx.y * TK;
  const int m_begin = blockIdx.z * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  if (m_begin >= m_end) return;

  constexpr int A_CH = TN / 8, B_CH = TK / 8;
  constexpr int A_CHUNKS = TN_BM * A_CH, B_CHUNKS = TN_BM * B_CH;
  constexpr int A_PER = (A_CHUNKS + 255) / 256, B_PER = (B_CHUNKS + 255) / 256;
  static_assert(A_CHUNKS % 256 == 0, "a thread must keep one column chunk of A (bias sums)");
  uint4 ra[A_PER], rb[B_PER];
  auto load_tiles = [&](int mb) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int c = tid + i * 256, r = c / A_CH, cc = c % A_CH;
      const int gm = mb + r, gn = n0 + cc * 8;
      ra[i] = make_uint4(0, 0, 0, 0);
      if (c < A_CHUNKS && gm < m_end && gn < N) ra[i] = *(const uint4*)(A + (size_t)gm * lda + gn);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int c = tid + i * 256, r = c / B_CH, cc = c % B_CH;
      const int gm = mb + r, gk = k0 + cc * 8;
      rb[i] = make_uint4(0, 0, 0, 0);
      if (c < B_CHUNKS && gm < m_end && gk < K) rb[i] = *(const uint4*)(B + (size_t)gm * ldb + gk);
    }
  };
  // fused bias gradient: column sums of A (= dY) ride along on the k-tile-0 blocks; a thread
  // always stages the same 8-column chunk (tid % A_CH), so it keeps 8 running sums
  const bool do_bias = (dbias != nullptr) && (blockIdx.y == 0);
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto store_tiles = [&](int buf) {
    unsigned char* la = lds[buf];
    unsigned char* lb = lds[buf] + TN_BM * ROWA;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int c = tid + i * 256;
      *(uint4*)(la + (c / A_CH) * ROWA + (c % A_CH) * 16) = ra[i];
      if (do_bias) {
        bsum[0] += lo_bf16(ra[i].x); bsum[1] += hi_bf16(ra[i].x);
        bsum[2] += lo_bf16(ra[i].y); bsum[3] += hi_bf16(ra[i].y);
        bsum[4] += lo_bf16(ra[i].z); bsum[5] += hi_bf16(ra[i].z);
        bsum[6] += lo_bf16(ra[i].w); bsum[7] += hi_bf16(ra[i].w);
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int c = tid + i * 256;
      if (c < B_CHUNKS) *(uint4*)(lb + (c / B_CH) * ROWB + (c % B_CH) * 16) = rb[i];
    }
  };

  f32x16_t acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read addressing: lane -> (half hh, column group cg, in-group i -> (q,p))
  const int hh = lane >> 5, cg = (lane >> 4) & 1, ii = lane & 15, q = ii >> 2, pp = ii & 3;
  const int a_off = (8 * hh + q) * ROWA + (wn * 64 + 16 * cg + 4 * pp) * 2;
  const int b_off = (8 * hh + q) * ROWB + (wk * 96 + 16 * cg + 4 * pp) * 2;

  const int nsteps = (m_end - m_begin + TN_BM - 1) / TN_BM;
  load_tiles(m_begin);
  store_tiles(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    if (s + 1 < nsteps) load_tiles(m_begin + (s + 1) * TN_BM);
    const unsigned char* la = lds[cur] + a_off;
    const unsigned char* lb = lds[cur] + TN_BM * ROWA + b_off;
#pragma unroll
    for (int ks = 0; ks < TN_BM / 16; ++ks) {
      bf16x8_t af[2], bfr[3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = make_bf16x8(lds_read_tr16(la + (ks * 16) * ROWA + i * 64),
                            lds_read_tr16(la + (ks * 16 + 4) * ROWA + i * 64));
#pragma unroll
      for (int j = 0; j < 3; ++j)
        bfr[j] = make_bf16x8(lds_read_tr16(lb + (ks * 16) * ROWB + j * 64),
                             lds_read_tr16(lb + (ks * 16 + 4) * ROWB + j * 64));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = mfma32(af[i], bfr[j], acc[i][j]);
    }
    if (s + 1 < nsteps) store_tiles(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = k0 + wk * 96 + j * 32 + (lane & 31);
      if (col >= K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = n0 + wn * 64 + i * 32 + acc_row(r, lane);
        if (row < N) atomicAdd(dW + (size_t)row * lddw + col, acc[i][j][r]);
      }
    }
  if (do_bias) {  // 256 / A_CH threads share a column chunk: reduce through LDS
    constexpr int GROUPS = 256 / A_CH;
    float* red = (float*)&lds[0][0];  // [GROUPS][TN]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid / A_CH) * TN + (tid % A_CH) * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < TN && n0 + tid < N) {
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < GROUPS; ++g) sum += red[g * TN + tid];
      atomicAdd(dbias + n0 + tid, sum);
    }
  }
}

__global__ void colsum_kernel(const bf16_t* __restrict__ A, int lda, float* __restrict__ out,
                              int M, int N, int rows_per_block) {
  // block (64 x 4): lane -> 2 adjacent columns (one dword), 4 row phases; grid.x = column
  // groups of 128, grid.y = row chunks
  const int col = (blockIdx.x * 64 + threadIdx.x) * 2;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s0 = 0.f, s1 = 0.f;
  if (col < N) {
    for (int r = r0 + threadIdx.y; r < r1; r += 4) {
      const uint32_t w = *(const uint32_t*)(A + (size_t)r * lda + col);
      s0 += lo_bf16(w);
      s1 += hi_bf16(w);
    }
  }
  __shared__ float red[4][64][2];
  red[threadIdx.y][threadIdx.x][0] = s0;
  red[threadIdx.y][threadIdx.x][1] = s1;
  __syncthreads();
  if (threadIdx.y == 0 && col < N) {
    for (int k = 1; k < 4; ++k) {
      s0 += red[k][threadIdx.x][0];
      s1 += red[k][threadIdx.x][1];
    }
    atomicAdd(out + col, s0);
    if (col + 1 < N) atomicAdd(out + col + 1, s1);
  }
}

}  // namespace

extern "C" int svit_gemm_tn(const void* A, int lda, const void* B, int ldb, float* dW, int lddw,
                            int M, int N, int K, int splits, float* dbias, void* stream) {
  if (!A || !B || !dW) return SVIT_ERR_ARG;
  if (M <= 0 || N <= 0 || K <= 0) return SVIT_ERR_SHAPE;
  if (lda % 8 != 0 || ldb % 8 != 0 || lda < N || ldb < K || lddw < K) return SVIT_ERR_ALIGN;
  if (((uintptr_t)A | (uintptr_t)B) & 15) return SVIT_ERR_ALIGN;
  // K >= 192: 128(n) x 192(k) tiles; narrower K: 256(n) x 96(k)
  const bool wide_k = K > 96;
  const int TN = wide_k ? 128 : 256, TK = wide_k ? 192 : 96;
  const int tiles = ((N + TN - 1) / TN) * ((K + TK - 1) / TK);
  if (splits <= 0) {
    // Cost model fitted on MI355X (tools/bench_kernels.py tnsplit): a block needs ~0.8 us per
    // 32-row step, 512 blocks run concurrently, and every split adds the whole [N,K] fp32
    // tile set with atomics at ~1.3 TB/s chip-wide.  Take the split count minimising the sum.
    const double atom_us = (double)N * K * 4.0 / 1.3e6;
    double best = 1e30;
    for (int s = 1; s <= 1024; s *= 2) {
      const long steps = ((M + s - 1) / s + TN_BM - 1) / TN_BM;
      if (steps < 2 && s > 1) break;
      const long rounds = ((long)tiles * s + 511) / 512;
      const double t = (double)rounds * steps * 0.8 + s * atom_us;
      if (t < best) { best = t; splits = s; }
    }
  }
  int rows_per_split = (M + splits - 1) / splits;
  rows_per_split = ((rows_per_split + TN_BM - 1) / TN_BM) * TN_BM;
  splits = (M + rows_per_split - 1) / rows_per_split;
  dim3 grid((N + TN - 1) / TN, (K + TK - 1) / TK, splits);
  if (wide_k)
    hipLaunchKernelGGL((gemm_tn_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)A, lda, (const bf16_t*)B, ldb, dW, lddw, M, N, K,
                       rows_per_split, dbias);
  else
    hipLaunchKernelGGL((gemm_tn_kernel<4, 1>), grid, dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)A, lda, (const bf16_t*)B, ldb, dW, lddw, M, N, K,
                       rows_per_split, dbias);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_colsum_bf16(const void* A, int lda, float* out, int M, int N, void* stream) {
  if (!A || !out || M <= 0 || N <= 0 || (N & 1) || (lda & 1)) return SVIT_ERR_ARG;
  const int col_groups = (N + 127) / 128;
  int row_chunks = 2048 / col_groups;
  if (row_chunks < 1) row_chunks = 1;
  int rows_per_block = (M + row_chunks - 1) / row_chunks;
  if (rows_per_block < 64) rows_per_block = 64;
  row_chunks = (M + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL(colsum_kernel, dim3(col_groups, row_chunks), dim3(64, 4), 0,
                     (hipStream_t)stream, (const bf16_t*)A, lda, out, M, N, rows_per_block);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}
