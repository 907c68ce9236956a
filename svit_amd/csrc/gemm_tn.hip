// bf16 MFMA weight-gradient GEMM for the nn.Linear family -- gfx950 only.
//   svit_gemm_tn : dW[N,K] += A[M,N]^T * B[M,K]  (reduction over rows, operands consumed
//                  through ds_read_b64_tr_b16 transposed LDS reads; fused bias gradient)
// (the forward / dgrad kernel svit_gemm_nt lives in gemm_nt.hip)
#include "common.h"
#include "../../include/svit_hip.h"

namespace {

// ---------------------------------------------------------------------------------------
// TN kernel: out tile 128(n) x 96(k), 4 waves, each wave one 32-row n-block x 96 k columns.
// Reduction rows arrive 32 at a time by LDS-DMA (global_load_lds_dwordx4) into a 4-stage ring
// behind a counted vmcnt and one raw barrier per step; both operands sit [m][cols] in LDS and
// are consumed through ds_read_b64_tr_b16 (inline asm: see attn_common.h on why not the
// builtin while DMA is in flight).  LDS images are lane-linear as the DMA requires:
//   A stage [32][128] bf16, 256-B rows; the four 64-B column groups of a row are XOR-swizzled
//     with (row & 3) on the SOURCE address so the 4 rows of a transposed block hit 4 bank groups;
//   B stage [32][96] bf16, 192-B rows: 4 consecutive rows already fall on disjoint bank groups.
// Rows past the split's end and columns past N / K are fetched from a zero block.
// The bias gradient (column sums of A) rides along as one extra MFMA against an all-ones
// operand on the k-tile-0 blocks.
// ---------------------------------------------------------------------------------------
constexpr int TN_BM = 32;        // reduction rows per stage
constexpr int TN_STAGES = 4;
constexpr int TN_TN = 128, TN_TK = 96;
constexpr int TN_A_BYTES = TN_BM * TN_TN * 2;   // 8 KB
constexpr int TN_B_BYTES = TN_BM * TN_TK * 2;   // 6 KB
constexpr int TN_STAGE_BYTES = TN_A_BYTES + TN_B_BYTES;
constexpr int TN_PER_WAVE = 4;   // 14 KB = 14 one-KB DMA instructions -> 4 per wave (2 repeats)

__device__ uint4 svit_zero16 = {0u, 0u, 0u, 0u};

template <int N>
__device__ __forceinline__ void tn_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A, int lda,
                                                         const bf16_t* __restrict__ B, int ldb,
                                                         float* __restrict__ dW, int lddw, int M,
                                                         int N, int K, int rows_per_split,
                                                         float* __restrict__ dbias) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[TN_STAGES * TN_STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * TN_TN, k0 = blockIdx.y * TN_TK;
  const int m_begin = blockIdx.z * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  if (m_begin >= m_end) return;
  const bf16_t* zero = (const bf16_t*)&svit_zero16;

  // DMA instruction n (0..13) of a stage: n < 8 -> A rows 4n..4n+3; else B chunks (n-8)*64..
  auto issue = [&](int step, int stage) {
    unsigned char* st = lds + stage * TN_STAGE_BYTES;
    const int mb = m_begin + step * TN_BM;
#pragma unroll
    for (int i = 0; i < TN_PER_WAVE; ++i) {
      int n = wave + i * 4;
      if (n > 13) n = 13;
      const bf16_t* src;
      unsigned char* dst;
      if (n < 8) {
        const int row = n * 4 + (lane >> 4), slot = lane & 15;
        const int grp = (slot >> 2) ^ (row & 3);             // logical 64-B column group
        const int col = n0 + grp * 32 + (slot & 3) * 8;
        src = (mb + row < m_end && col < N) ? A + (size_t)(mb + row) * lda + col : zero;
        dst = st + n * 1024;
      } else {
        const int q = (n - 8) * 64 + lane;                   // chunk index in the B stage
        const int row = q / 12, col = k0 + (q % 12) * 8;
        src = (mb + row < m_end && col < K) ? B + (size_t)(mb + row) * ldb + col : zero;
        dst = st + TN_A_BYTES + (n - 8) * 1024;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  f32x16_t acc[3], accb;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; acc[2][r] = 0.f; accb[r] = 0.f; }
  const bool do_bias = (dbias != nullptr) && (blockIdx.y == 0);
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  // transposed-read addresses within a stage (k-step 0); lane -> (hh, cg, q, pp)
  const int hh = lane >> 5, cg = (lane >> 4) & 1, ii = lane & 15, q = ii >> 2, pp = ii & 3;
  const unsigned lds0 = (unsigned)(size_t)lds;
  const unsigned a_off = (8 * hh + q) * 256 + ((wave ^ q) & 3) * 64 + 32 * cg + 8 * pp;
  const unsigned b_off = TN_A_BYTES + (8 * hh + q) * 192 + 32 * cg + 8 * pp;

  const int nsteps = (m_end - m_begin + TN_BM - 1) / TN_BM;
#pragma unroll
  for (int s = 0; s < TN_STAGES - 1; ++s)
    if (s < nsteps) issue(s, s);
  for (int s = 0; s < nsteps; ++s) {
    if (s + TN_STAGES - 2 < nsteps) tn_wait_vmcnt<(TN_STAGES - 2) * TN_PER_WAVE>();
    else tn_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (s + TN_STAGES - 1 < nsteps) issue(s + TN_STAGES - 1, (s + TN_STAGES - 1) % TN_STAGES);
    const unsigned st = lds0 + (s % TN_STAGES) * TN_STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      s16x4_t al, ah, b0l, b0h, b1l, b1h, b2l, b2h;
      const unsigned pa = st + a_off + ks * 16 * 256, pb = st + b_off + ks * 16 * 192;
      asm volatile(
          "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:1024\n\t"
          "ds_read_b64_tr_b16 %2, %9\n\tds_read_b64_tr_b16 %3, %9 offset:768\n\t"
          "ds_read_b64_tr_b16 %4, %9 offset:64\n\tds_read_b64_tr_b16 %5, %9 offset:832\n\t"
          "ds_read_b64_tr_b16 %6, %9 offset:128\n\tds_read_b64_tr_b16 %7, %9 offset:896\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(al), "=&v"(ah), "=&v"(b0l), "=&v"(b0h), "=&v"(b1l), "=&v"(b1h), "=&v"(b2l), "=&v"(b2h)
          : "v"(pa), "v"(pb)
          : "memory");
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8_t af = make_bf16x8(al, ah);
      acc[0] = mfma32(af, make_bf16x8(b0l, b0h), acc[0]);
      acc[1] = mfma32(af, make_bf16x8(b1l, b1h), acc[1]);
      acc[2] = mfma32(af, make_bf16x8(b2l, b2h), acc[2]);
      if (do_bias) accb = mfma32(af, ones, accb);
    }
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
  if (do_bias && (lane & 31) == 0) {   // every column of accb holds the same column sums
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = n0 + wave * 32 + acc_row(r, lane);
      if (row < N) atomicAdd(dbias + row, accb[r]);
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
