// bf16 MFMA GEMMs for the nn.Linear family -- gfx950 only.
//   svit_gemm_nt : C[M,N] = A[M,K] * W[N,K]^T, fused epilogues (bias / GELU / residual +
//                  DropPath / fp32 accumulate / GELU-backward)
//   svit_gemm_tn : dW[N,K] += A[M,N]^T * B[M,K]  (weight gradients, reduction over rows,
//                  operands consumed through ds_read_b64_tr_b16 transposed LDS reads)
// Tiling is wave64-native: every wave owns (WM x 96) of the output as 32x32x16 MFMA
// accumulators; A/W tiles are register-staged into padded (bank-conflict-free) LDS rows,
// double-buffered, one barrier per K-step.
#include "gemm_epilogue.h"

namespace {

template <int BK> struct LdsRow { static constexpr int kBytes = BK * 2 + 16; };

// ---------------------------------------------------------------------------------------
// NT kernel.  Block = WAVES_M x WAVES_N waves; wave tile = (32*RB) x 96.
// ---------------------------------------------------------------------------------------
template <int RB, int NB, int WAVES_M, int WAVES_N, int BK, int EPI>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 2) void gemm_nt_kernel(svit_gemm_args p) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = 32 * RB * WAVES_M;
  constexpr int WN = 32 * NB;               // columns per wave
  constexpr int BN = WN * WAVES_N;
  constexpr int EP_LD = WN + 4;             // fp32 row stride of the epilogue staging block
  constexpr int ROWB = LdsRow<BK>::kBytes;
  constexpr int CH = BK / 8;                 // 16-byte chunks per tile row
  constexpr int A_CHUNKS = BM * CH, W_CHUNKS = BN * CH;
  constexpr int A_PER = (A_CHUNKS + NT - 1) / NT, W_PER = (W_CHUNKS + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STAGE = (BM + BN) * ROWB;   // [A tile | W tile] per pipeline stage

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // Tile order: N tiles fastest, so consecutive tiles share the A row panel.  The dispatcher
  // deals consecutive workgroups round-robin over the 8 XCDs (private L2 each), so remap the
  // linear id to give every XCD a CONTIGUOUS run of tiles (bijective for any grid size).
  const int nwg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
  const int wgid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
  const int m0 = (wgid / gridDim.x) * BM, n0 = (wgid % gridDim.x) * BN;
  const bf16_t* A = (const bf16_t*)p.A;
  const bf16_t* W = (const bf16_t*)p.W;

  uint4 ra[A_PER], rw[W_PER];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int c = tid + i * NT;
      const int r = c / CH, cc = c % CH;
      const int gr = m0 + r;
      ra[i] = make_uint4(0, 0, 0, 0);
      if (c < A_CHUNKS && gr < p.M)
        ra[i] = *(const uint4*)(A + (size_t)gr * p.lda + k0 + cc * 8);
    }
#pragma unroll
    for (int i = 0; i < W_PER; ++i) {
      const int c = tid + i * NT;
      const int r = c / CH, cc = c % CH;
      const int gr = n0 + r;
      rw[i] = make_uint4(0, 0, 0, 0);
      if (c < W_CHUNKS && gr < p.N)
        rw[i] = *(const uint4*)(W + (size_t)gr * p.ldw + k0 + cc * 8);
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int c = tid + i * NT;
      if (c < A_CHUNKS) *(uint4*)(smem + buf * STAGE + (c / CH) * ROWB + (c % CH) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < W_PER; ++i) {
      const int c = tid + i * NT;
      if (c < W_CHUNKS)
        *(uint4*)(smem + buf * STAGE + BM * ROWB + (c / CH) * ROWB + (c % CH) * 16) = rw[i];
    }
  };

  f32x16_t acc[RB][NB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  const int frag_off = (lane & 31) * ROWB + (lane >> 5) * 16;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles((kt + 1) * BK);
    const unsigned char* la = smem + cur * STAGE + (wm * 32 * RB) * ROWB + frag_off;
    const unsigned char* lw = smem + cur * STAGE + BM * ROWB + (wn * WN) * ROWB + frag_off;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8_t af[RB], wf[NB];
#pragma unroll
      for (int i = 0; i < RB; ++i) af[i] = *(const bf16x8_t*)(la + i * 32 * ROWB + ks * 32);
#pragma unroll
      for (int j = 0; j < NB; ++j) wf[j] = *(const bf16x8_t*)(lw + j * 32 * ROWB + ks * 32);
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = mfma32(af[i], wf[j], acc[i][j]);
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

  nt_epilogue<RB, NB, EPI>(p, acc, smem, m0, n0, wm, wn, lane, wave);
}

template <int RB, int NB, int WAVES_M, int WAVES_N, int BK>
int launch_nt(const svit_gemm_args& a, hipStream_t st) {
  constexpr int BM = 32 * RB * WAVES_M, BN = 32 * NB * WAVES_N;
  constexpr int NT = WAVES_M * WAVES_N * 64;
  size_t lds = 2 * (size_t)(BM + BN) * LdsRow<BK>::kBytes;
  const size_t lds_epi = (size_t)WAVES_M * WAVES_N * 16 * (32 * NB + 4) * sizeof(float);
  if (lds < lds_epi) lds = lds_epi;
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM);
#define SVIT_NT_CASE(E)                                                                  \
  case E:                                                                                \
    hipLaunchKernelGGL((gemm_nt_kernel<RB, NB, WAVES_M, WAVES_N, BK, E>), grid, dim3(NT), lds, st, a); \
    break;
  switch (a.epilogue) {
    SVIT_NT_CASE(SVIT_EPI_BF16)
    SVIT_NT_CASE(SVIT_EPI_GELU)
    SVIT_NT_CASE(SVIT_EPI_RESID)
    SVIT_NT_CASE(SVIT_EPI_F32)
    SVIT_NT_CASE(SVIT_EPI_DGELU)
    default:
      return SVIT_ERR_ARG;
  }
#undef SVIT_NT_CASE
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

// ---------------------------------------------------------------------------------------
// TN kernel: out tile 128(n) x 96(k), 4 waves, each wave one 32-row n-block x 96 k columns.
// Both operands are [rows=m][cols] in LDS and read transposed (ds_read_b64_tr_b16).
// ---------------------------------------------------------------------------------------
constexpr int TN_BM = 64;        // reduction rows per step
constexpr int TN_TN = 128, TN_TK = 96;
constexpr int TN_ROWA = TN_TN * 2 + 64;  // 320 B: the 4 rows of a tr block hit disjoint banks
constexpr int TN_ROWB = TN_TK * 2;       // 192 B: conflict-free as is

__global__ __launch_bounds__(256) void gemm_tn_kernel(const bf16_t* __restrict__ A, int lda,
                                                      const bf16_t* __restrict__ B, int ldb,
                                                      float* __restrict__ dW, int lddw, int M,
                                                      int N, int K, int rows_per_split,
                                                      float* __restrict__ dbias) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][TN_BM * (TN_ROWA + TN_ROWB)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * TN_TN, k0 = blockIdx.y * TN_TK;
  const int m_begin = blockIdx.z * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
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
  const bool do_bias = (dbias != nullptr) && (blockIdx.y == 0);
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

extern "C" int svit_gemm_nt(const svit_gemm_args* args, void* stream) {
  if (!args || !args->A || !args->W || !args->out) return SVIT_ERR_ARG;
  const svit_gemm_args& a = *args;
  if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.K % 32 != 0 || a.N % 96 != 0) return SVIT_ERR_SHAPE;
  if (a.lda % 8 != 0 || a.ldw % 8 != 0 || a.lda < a.K || a.ldw < a.K || a.ldo < a.N)
    return SVIT_ERR_ALIGN;
  if (((uintptr_t)a.A | (uintptr_t)a.W) & 15) return SVIT_ERR_ALIGN;
  if (a.ldo % 4 != 0 || ((uintptr_t)a.out & 15)) return SVIT_ERR_ALIGN;
  if (a.bias && ((uintptr_t)a.bias & 15)) return SVIT_ERR_ALIGN;
  if (a.aux && (a.ldaux % 4 != 0 || ((uintptr_t)a.aux & 15))) return SVIT_ERR_ALIGN;
  if (a.out2 && (a.ldo2 % 4 != 0 || ((uintptr_t)a.out2 & 15))) return SVIT_ERR_ALIGN;
  if (a.epilogue == SVIT_EPI_GELU && !a.out2) return SVIT_ERR_ARG;
  if ((a.epilogue == SVIT_EPI_RESID || a.epilogue == SVIT_EPI_DGELU) && !a.aux) return SVIT_ERR_ARG;
  if (a.epilogue == SVIT_EPI_RESID && a.row_scale && a.rows_per_sample <= 0) return SVIT_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  // Tile choice: arithmetic intensity (FLOP per L2 byte) grows with the tile, so take the
  // 256x192 tile (4 waves x (64 x 192)) whenever it still yields >= ~1.5 tiles per CU;
  // otherwise 128x192.  N == 96 (mod 192): 256x96 for tall problems, 128x96 for the rest.
  static bool configured = false;
  if (!configured) {   // the 256x192 pipeline needs > 64 KB of dynamic LDS
#define SVIT_SET_LDS(E) hipFuncSetAttribute((const void*)gemm_nt_kernel<2, 6, 4, 1, 32, E>, \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)
    SVIT_SET_LDS(SVIT_EPI_BF16); SVIT_SET_LDS(SVIT_EPI_GELU); SVIT_SET_LDS(SVIT_EPI_RESID);
    SVIT_SET_LDS(SVIT_EPI_F32); SVIT_SET_LDS(SVIT_EPI_DGELU);
#undef SVIT_SET_LDS
    configured = true;
  }
  if (a.N % 192 == 0) {
    const long tiles_big = (long)((a.M + 255) / 256) * (a.N / 192);
    if (tiles_big >= 384) return launch_nt<2, 6, 4, 1, 32>(a, st);
    return launch_nt<2, 3, 2, 2, 32>(a, st);
  }
  if (a.M >= 8192) return launch_nt<2, 3, 4, 1, 32>(a, st);
  return launch_nt<1, 3, 4, 1, 32>(a, st);
}

extern "C" int svit_gemm_tn(const void* A, int lda, const void* B, int ldb, float* dW, int lddw,
                            int M, int N, int K, int splits, float* dbias, void* stream) {
  if (!A || !B || !dW) return SVIT_ERR_ARG;
  if (M <= 0 || N <= 0 || K <= 0) return SVIT_ERR_SHAPE;
  if (lda % 8 != 0 || ldb % 8 != 0 || lda < N || ldb < K || lddw < K) return SVIT_ERR_ALIGN;
  if (((uintptr_t)A | (uintptr_t)B) & 15) return SVIT_ERR_ALIGN;
  const int tiles = ((N + TN_TN - 1) / TN_TN) * ((K + TN_TK - 1) / TN_TK);
  if (splits <= 0) {
    // every split adds the whole [N,K] tile set with fp32 atomics (~1.3 TB/s chip-wide): take
    // just enough splits to fill the chip (~2 blocks per CU), at least 4 reduction steps each
    splits = (512 + tiles - 1) / tiles;
    const int max_by_rows = (M + 4 * TN_BM - 1) / (4 * TN_BM);
    if (splits > max_by_rows) splits = max_by_rows;
    if (splits < 1) splits = 1;
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
