// bf16 MFMA weight-gradient GEMM for the nn.Linear family -- gfx950 only.
//   svit_gemm_tn : dW[N,K] += A[M,N]^T * B[M,K]  (reduction over rows, operands consumed
//                  through ds_read_b64_tr_b16 transposed LDS reads; fused bias gradient)
// (the forward / dgrad kernel svit_gemm_nt lives in gemm_nt.hip)
#include <atomic>
#include "common.h"
#include "../../include/svit_hip.h"

namespace {

// ---------------------------------------------------------------------------------------
// TN kernel: out tile 128(n) x 96(k), 4 waves, each wave one 32-row n-block x 96 k columns.
// Both operands are [rows=m][cols] in LDS and read transposed (ds_read_b64_tr_b16).
// ---------------------------------------------------------------------------------------
constexpr int TN_BM = 64;        // reduction rows per step
constexpr int TN_TN = 128, TN_TK = 96;
constexpr int TN_ROWA = TN_TN * 2 + 64;  // 320 B: the 4 rows of a tr block hit disjoint banks
constexpr int TN_ROWB = TN_TK * 2;       // 192 B: conflict-free as is

typedef unsigned char tn_lds_t[2][TN_BM * (TN_ROWA + TN_ROWB)];

// One workgroup's share: rows [m_begin, m_end) of the reduction for the 128x96 tile at (n0, k0).
__device__ __forceinline__ void tn_tile(tn_lds_t& lds, const bf16_t* __restrict__ A, int lda,
                                        const bf16_t* __restrict__ B, int ldb,
                                        float* __restrict__ dW, int lddw, int N, int K, int n0,
                                        int k0, int m_begin, int m_end,
                                        float* __restrict__ dbias, bool bias_tile) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (m_begin >= m_end) return;

  constexpr int A_CH = TN_TN / 8, B_CH = TN_TK / 8;        // 16 / 12 chunks per row
  constexpr int A_PER = TN_BM * A_CH / 256, B_PER = TN_BM * B_CH / 256;  // 4 / 3
  uint4 ra[A_PER], rb[B_PER];
  auto load_tiles = [&](int mb) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int c = tid + i * 256, r = c / A_CH, cc = c % A_CH;
      const int gm = mb + r, gn = n0 + cc * 8;
      ra[i] = make_uint4(0, 0, 0, 0);
      if (gm < m_end && gn < N) ra[i] = *(const uint4*)(A + (size_t)gm * lda + gn);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int c = tid + i * 256, r = c / B_CH, cc = c % B_CH;
      const int gm = mb + r, gk = k0 + cc * 8;
      rb[i] = make_uint4(0, 0, 0, 0);
      if (gm < m_end && gk < K) rb[i] = *(const uint4*)(B + (size_t)gm * ldb + gk);
    }
  };
  // fused bias gradient: column sums of A (= dY) ride along on the k-tile-0 blocks; a thread
  // always stages the same 8-column chunk (tid % 16), so it keeps 8 running sums
  const bool do_bias = (dbias != nullptr) && bias_tile;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto store_tiles = [&](int buf) {
    unsigned char* la = lds[buf];
    unsigned char* lb = lds[buf] + TN_BM * TN_ROWA;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int c = tid + i * 256;
      *(uint4*)(la + (c / A_CH) * TN_ROWA + (c % A_CH) * 16) = ra[i];
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
      *(uint4*)(lb + (c / B_CH) * TN_ROWB + (c % B_CH) * 16) = rb[i];
    }
  };

  f32x16_t acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // transposed-read addressing: lane -> (half hh, column group cg, in-group i -> (q,p))
  const int hh = lane >> 5, cg = (lane >> 4) & 1, ii = lane & 15, q = ii >> 2, pp = ii & 3;
  const int a_off = (8 * hh + q) * TN_ROWA + (wave * 32 + 16 * cg + 4 * pp) * 2;
  const int b_off = (8 * hh + q) * TN_ROWB + (16 * cg + 4 * pp) * 2;

  const int nsteps = (m_end - m_begin + TN_BM - 1) / TN_BM;
  load_tiles(m_begin);
  store_tiles(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    if (s + 1 < nsteps) load_tiles(m_begin + (s + 1) * TN_BM);
    const unsigned char* la = lds[cur] + a_off;
    const unsigned char* lb = lds[cur] + TN_BM * TN_ROWA + b_off;
#pragma unroll
    for (int ks = 0; ks < TN_BM / 16; ++ks) {
      const bf16x8_t af = make_bf16x8(lds_read_tr16(la + (ks * 16) * TN_ROWA),
                                      lds_read_tr16(la + (ks * 16 + 4) * TN_ROWA));
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const bf16x8_t bfr = make_bf16x8(lds_read_tr16(lb + (ks * 16) * TN_ROWB + j * 64),
                                         lds_read_tr16(lb + (ks * 16 + 4) * TN_ROWB + j * 64));
        acc[j] = mfma32(af, bfr, acc[j]);
      }
    }
    if (s + 1 < nsteps) store_tiles(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int col = k0 + j * 32 + (lane & 31);
    if (col >= K) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = n0 + wave * 32 + acc_row(r, lane);
      if (row < N) atomicAdd(dW + (size_t)row * lddw + col, acc[j][r]);
    }
  }
  if (do_bias) {  // 16 threads share a column chunk: reduce through LDS, one atomic per column
    float* red = (float*)&lds[0][0];  // [16][128]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid / A_CH) * TN_TN + (tid % A_CH) * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < TN_TN && n0 + tid < N) {
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) sum += red[g * TN_TN + tid];
      atomicAdd(dbias + n0 + tid, sum);
    }
  }
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(const bf16_t* __restrict__ A, int lda,
                                                      const bf16_t* __restrict__ B, int ldb,
                                                      float* __restrict__ dW, int lddw, int M,
                                                      int N, int K, int rows_per_split,
                                                      float* __restrict__ dbias) {
  __shared__ __attribute__((aligned(16))) tn_lds_t lds;
  const int m_begin = blockIdx.z * rows_per_split;
  tn_tile(lds, A, lda, B, ldb, dW, lddw, N, K, blockIdx.x * TN_TN, blockIdx.y * TN_TK, m_begin,
          min(M, m_begin + rows_per_split), dbias, blockIdx.y == 0);
}

// Grouped launch: up to SVIT_TN_GROUP_MAX independent weight-gradient GEMMs share one grid, so
// the machine-wide accumulator flush (every resident workgroup atomically adds its 128x96 fp32
// tile: ~25 MB per launch, the dominant cost of a short-reduction wgrad) is paid once per group
// instead of once per GEMM.
struct TnGroup {
  svit_tn_problem p[SVIT_TN_GROUP_MAX];
  int first_block[SVIT_TN_GROUP_MAX + 1];
  int tiles_n[SVIT_TN_GROUP_MAX];
  int tiles[SVIT_TN_GROUP_MAX];
  int rows_per_split[SVIT_TN_GROUP_MAX];
  int count;
};

__global__ __launch_bounds__(256) void gemm_tn_grouped_kernel(const TnGroup g) {
  __shared__ __attribute__((aligned(16))) tn_lds_t lds;
  // Logical ids are (problem, split)-major, tile-minor: the tiles of one split walk the SAME
  // rows in lock-step and re-read each other's A / B column panels (A once per k-tile, B once
  // per n-tile).  Consecutive logical ids are therefore placed on ONE XCD, so those re-reads
  // hit its L2 instead of crossing the fabric (rocprofv3 FETCH_SIZE before: 810 MB per launch
  // against ~230 MB of operands -- the kernel ran at HBM speed).
  const int nwg = gridDim.x, lin = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
  const int bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < SVIT_TN_GROUP_MAX; ++i)
    if (i < g.count && bid >= g.first_block[i]) pi = i;
  const svit_tn_problem& p = g.p[pi];
  const int local = bid - g.first_block[pi];
  const int tile = local % g.tiles[pi], split = local / g.tiles[pi];
  const int tn = tile % g.tiles_n[pi], tk = tile / g.tiles_n[pi];
  const int m_begin = split * g.rows_per_split[pi];
  tn_tile(lds, (const bf16_t*)p.A, p.lda, (const bf16_t*)p.B, p.ldb, p.dW, p.lddw, p.N, p.K,
          tn * TN_TN, tk * TN_TK, m_begin, min(p.M, m_begin + g.rows_per_split[pi]), p.dbias,
          tk == 0);
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
  const int tiles = ((N + TN_TN - 1) / TN_TN) * ((K + TN_TK - 1) / TN_TK);
  if (splits <= 0) {
    // Cost model fitted on MI355X (tools/bench_kernels.py tnsplit): a block needs ~0.85 us per
    // 64-row step, 512 blocks run concurrently, and every split adds the whole [N,K] fp32 tile
    // set with atomics at ~0.75 TB/s effective.  Take the split count minimising the sum.
    const double atom_us = (double)N * K * 4.0 / 0.75e6;
    double best = 1e30;
    static const int cand[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512};
    for (int ci = 0; ci < (int)(sizeof(cand) / sizeof(cand[0])); ++ci) {
      const int s = cand[ci];
      const long steps = ((M + s - 1) / s + TN_BM - 1) / TN_BM;
      if (steps < 2 && s > 1) break;
      const long rounds = ((long)tiles * s + 511) / 512;
      const double t = (double)rounds * steps * 0.85 + s * atom_us;
      if (t < best) { best = t; splits = s; }
    }
  }
  int rows_per_split = (M + splits - 1) / splits;
  rows_per_split = ((rows_per_split + TN_BM - 1) / TN_BM) * TN_BM;
  splits = (M + rows_per_split - 1) / rows_per_split;
  dim3 grid((N + TN_TN - 1) / TN_TN, (K + TN_TK - 1) / TN_TK, splits);
  hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)A,
                     lda, (const bf16_t*)B, ldb, dW, lddw, M, N, K, rows_per_split, dbias);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

// cost-model constants of the grouped launch (svit_debug_set keys 2 / 3 for sweeps)
static std::atomic<double> g_tn_step_us{0.85};      // one 64-row step of a workgroup, 512 resident
static std::atomic<double> g_tn_atomic_tbs{0.75};   // effective fp32 atomic flush rate, TB/s
extern "C" int svit_debug_set_tn(int step_us_x100, int atomic_tbs_x100) {
  if (step_us_x100 > 0) g_tn_step_us = step_us_x100 * 0.01;
  if (atomic_tbs_x100 > 0) g_tn_atomic_tbs = atomic_tbs_x100 * 0.01;
  return SVIT_OK;
}

static int tn_check(const svit_tn_problem& p) {
  if (!p.A || !p.B || !p.dW) return SVIT_ERR_ARG;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return SVIT_ERR_SHAPE;
  if (p.lda % 8 != 0 || p.ldb % 8 != 0 || p.lda < p.N || p.ldb < p.K || p.lddw < p.K)
    return SVIT_ERR_ALIGN;
  if (((uintptr_t)p.A | (uintptr_t)p.B) & 15) return SVIT_ERR_ALIGN;
  return SVIT_OK;
}

static int tn_grouped(const svit_tn_problem* probs, int count, int ordered, void* stream);

extern "C" int svit_gemm_tn_grouped(const svit_tn_problem* probs, int count, void* stream) {
  return tn_grouped(probs, count, 0, stream);
}

// ordered != 0: no problem is cut along its reduction rows, so every element of every dW
// receives exactly ONE atomic add -- the weight gradients are bit-reproducible (a debugging /
// regression-diff mode: few workgroups, several times slower than the split form).
extern "C" int svit_gemm_tn_grouped_ex(const svit_tn_problem* probs, int count, int ordered,
                                       void* stream) {
  return tn_grouped(probs, count, ordered, stream);
}

static int tn_grouped(const svit_tn_problem* probs, int count, int ordered, void* stream) {
  if (!probs || count <= 0) return SVIT_ERR_ARG;
  for (int i = 0; i < count; ++i) {
    const int rc = tn_check(probs[i]);
    if (rc) return rc;
  }
  for (int base = 0; base < count; base += SVIT_TN_GROUP_MAX) {
    TnGroup g;
    g.count = count - base < SVIT_TN_GROUP_MAX ? count - base : SVIT_TN_GROUP_MAX;
    long max_steps = 1;
    for (int i = 0; i < g.count; ++i) {
      g.p[i] = probs[base + i];
      g.tiles_n[i] = (g.p[i].N + TN_TN - 1) / TN_TN;
      g.tiles[i] = g.tiles_n[i] * ((g.p[i].K + TN_TK - 1) / TN_TK);
      const long st = (g.p[i].M + TN_BM - 1) / TN_BM;
      if (st > max_steps) max_steps = st;
    }
    // Every problem is cut into chunks of `steps` 64-row steps, so all workgroups run equally
    // long.  Same fitted model as svit_gemm_tn: 0.85 us per step with 512 resident workgroups,
    // plus the atomic flush of one 128x96 fp32 tile per workgroup at ~0.75 TB/s.
    double best = 1e30;
    long best_steps = max_steps;
    for (long steps = 2; steps <= max_steps; steps += (steps < 32 ? 1 : steps / 16)) {
      long blocks = 0;
      for (int i = 0; i < g.count; ++i) {
        const long st = (g.p[i].M + TN_BM - 1) / TN_BM;
        blocks += (long)g.tiles[i] * ((st + steps - 1) / steps);
      }
      const double t = (double)((blocks + 511) / 512) * steps * g_tn_step_us +
                       (double)blocks * (TN_TN * TN_TK * 4.0) / (g_tn_atomic_tbs * 1e6);
      if (t < best) { best = t; best_steps = steps; }
    }
    if (ordered) best_steps = max_steps;
    int total = 0;
    for (int i = 0; i < g.count; ++i) {
      g.rows_per_split[i] = (int)best_steps * TN_BM;
      const int splits = (g.p[i].M + g.rows_per_split[i] - 1) / g.rows_per_split[i];
      g.first_block[i] = total;
      total += g.tiles[i] * splits;
    }
    for (int i = g.count; i <= SVIT_TN_GROUP_MAX; ++i) g.first_block[i] = total;
    hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, g);
    SVIT_LAUNCH_CHECK();
  }
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
