// bf16 MFMA weight-gradient GEMM for the nn.Linear family -- gfx950 only.
//   svit_gemm_tn : dW[N,K] += A[M,N]^T * B[M,K]  (reduction over rows, operands consumed
//                  through ds_read_b64_tr_b16 transposed LDS reads; fused bias gradient)
// (the forward / dgrad kernel svit_gemm_nt lives in gemm_nt.hip)
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "../../include/svit_hip.h"

namespace {

template <int B, int E, typename F>
__device__ __forceinline__ void tn_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    tn_static_for<B + 1, E>(f);
  }
}

// ---------------------------------------------------------------------------------------
// TN kernel.  Both operands are [rows=m][cols] in LDS and read transposed (ds_read_b64_tr_b16).
//   TnCfg<1, 4, 1, 64>: out tile 128(n) x 96(k), 4 waves of 32 x 96, 64 reduction rows per step.
//   TnCfg<2, 2, 2, 32> (round 2): out tile 128 x 192, 2 x 2 waves of 64 x 96, 32 rows per step --
//     the same 12 MFMAs per wave and step from 20 KB of staged operands instead of 28 KB
//     (rocprofv3 round 2: the grouped launch fetched 687 MB against ~320 MB of operands and the
//     CU's vector-memory path, not the matrix pipe, sets the pace), and 10 transposed LDS reads
//     per 6 MFMAs instead of 8 per 3.
// ---------------------------------------------------------------------------------------
template <int RB_, int WN_, int WK_, int BM_>
struct TnCfg {
  static constexpr int RB = RB_, WN = WN_, WK = WK_, BM = BM_;
  static constexpr int TN = 32 * RB * WN, TK = 96 * WK;
  static constexpr int ROWA = TN * 2 + 64;                      // 320 B: the 4 rows of a tr block hit disjoint banks
  static constexpr int ROWB = TK * 2 + ((TK * 2) % 256 == 128 ? 64 : 0);   // 192 B as is; 384 -> 448
  static constexpr int STAGE = BM * (ROWA + ROWB);
};
using TnSmall = TnCfg<1, 4, 1, 64>;
using TnBig = TnCfg<2, 2, 2, 32>;

// ---------------------------------------------------------------------------------------
// One workgroup's share: rows [m_begin, m_end) of the reduction for the TN x TK tile at (n0, k0).
// Round 3: through an LDS-DMA ring.  The round-2 body (git history) staged both operands
// through registers (global_load -> VGPR -> ds_write_b128: 20-28 KB of LDS stores per 12 MFMAs,
// half of them behind bounds-check branches) and takes one __syncthreads per 12 MFMAs with the
// transposed reads left to the compiler.  Here:
//   * A / B stages [m][cols] arrive by buffer-descriptor LDS-DMA (no VGPR staging, no ds_write, no
//     per-piece address arithmetic: per-lane byte offsets are stage-invariant, the stage's first
//     row is the scalar offset), NS stages deep behind a counted vmcnt and ONE raw barrier per stage;
//   * rows are unpadded (a DMA writes 1 KB linearly); the 4 rows x 64 B of a transposed read land on
//     disjoint banks through a swizzle of the 64-byte blocks on the per-lane SOURCE address
//     (A, 256-B rows: block ^ (row & 3); B, 384-B rows: block ^ ((row >> 1) & 1); 192-B rows skew by
//     themselves) and the same swizzle in the read address;
//   * the transposed fragment reads of k-step s+1 are issued (inline asm) before the MFMAs of k-step
//     s and released by counted lgkmcnt waits;
//   * a ragged last stage (rows past m_end) is fetched through clamped rows and its A rows are
//     zeroed in LDS; columns past N / K are clamped in the source address (their products are
//     never stored).
template <int RB_, int WN_, int WK_, int BM_, int NS_>
struct TnDma {
  static constexpr int RB = RB_, WN = WN_, WK = WK_, BM = BM_, NS = NS_;
  static constexpr int TN = 32 * RB * WN, TK = 96 * WK;
  static constexpr int ROWA = TN * 2, ROWB = TK * 2;             // 256 B; 192 / 384 B
  static constexpr int A_BYTES = BM * ROWA, B_BYTES = BM * ROWB, STAGE = A_BYTES + B_BYTES;
  static constexpr int A_INSTR = A_BYTES / 1024, B_INSTR = B_BYTES / 1024;
  static constexpr int NW = WN * WK, NT = NW * 64;               // waves / threads of a workgroup
  static constexpr int PER = (A_INSTR + B_INSTR) / NW;           // DMA instructions per wave and stage
  static_assert((A_INSTR + B_INSTR) % NW == 0, "every wave issues the same number of pieces");
  __host__ __device__ static constexpr int swz_a(int row) { return row & 3; }
  // 384-byte rows alternate between two bank offsets (one swizzle bit), rows that are a multiple of 256 bytes all
  // start on the same bank (two bits, like A), 192-byte rows skew by themselves
  __host__ __device__ static constexpr int swz_b(int row) { return ROWB == 384 ? ((row >> 1) & 1) : (ROWB % 256 == 0 ? (row & 3) : 0); }
};
using TnSmallD = TnDma<1, 4, 1, 64, 2>;     // 128 x 96 tiles, 64 rows per stage, 2 stages (56 KB)
#ifndef SVIT_TN_BIG_NS      // ring depth of the 128 x 192 tile (diagnostic builds: 4 = 80 KB, still two workgroups per CU)
#define SVIT_TN_BIG_NS 3
#endif
#ifndef SVIT_TN_BIG_BM      // reduction rows per stage of the 128 x 192 tile (diagnostic builds: 64 with 2 stages = half the barriers)
#define SVIT_TN_BIG_BM 32
#endif
using TnBigD = TnDma<2, 2, 2, SVIT_TN_BIG_BM, SVIT_TN_BIG_NS>;       // 128 x 192 tiles, 32 rows per stage, 3 stages (60 KB)
constexpr int TN_DMA_LDS = 2 * TnSmallD::STAGE > SVIT_TN_BIG_NS * TnBigD::STAGE ? 2 * TnSmallD::STAGE : SVIT_TN_BIG_NS * TnBigD::STAGE;

template <int OFF>
__device__ __forceinline__ void tn_read_tr(s16x4_t& d, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "i"(OFF) : "memory");
}

template <class C>
__device__ __forceinline__ void tn_tile_dma(unsigned char* lds, const bf16_t* __restrict__ A, int lda,
                                            const bf16_t* __restrict__ B, int ldb,
                                            float* __restrict__ dW, int lddw, int M, int N, int K, int n0,
                                            int k0, int m_begin, int m_end,
                                            float* __restrict__ dbias, bool bias_tile) {
#if __HIP_DEVICE_COMPILE__
  constexpr int RB = C::RB, BM = C::BM, NS = C::NS, PER = C::PER;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave / C::WK, wk = wave % C::WK;
  if (m_begin >= m_end) return;
  // ---- per-lane source offsets of this wave's DMA pieces (stage-invariant)
  // (a wave's first A_INSTR / 4 pieces are A pieces wave, wave + 4, ...; the rest B pieces)
  constexpr int NW = C::NW, NT = C::NT;
  constexpr int PA = C::A_INSTR / NW;
  static_assert(C::A_INSTR % NW == 0 && C::B_INSTR % NW == 0, "pieces split evenly over the waves");
  unsigned voff[PER];
  const int na8 = (N - n0 + 7) / 8, nb8 = (K - k0 + 7) / 8;     // valid 16-byte column chunks of this tile
  auto piece = [&](int i, bool a) { return wave + NW * (a ? i : i - PA); };
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const bool a = i < PA;
    const int o = piece(i, a) * 1024 + lane * 16;                // byte offset inside the A / B image
    const int rowb = a ? C::ROWA : C::ROWB;
    const int row = o / rowb, inrow = o % rowb;
    const int pb = inrow >> 6, c16 = (inrow >> 4) & 3;
    const int lb = pb ^ (a ? C::swz_a(row) : C::swz_b(row));     // logical 64-byte block
    const int ch = min(lb * 4 + c16, (a ? na8 : nb8) - 1);       // logical 16-byte chunk, clamped into the matrix
    voff[i] = (unsigned)(row * (a ? lda : ldb) + (a ? n0 : k0) + ch * 8) * 2u;
  }
  const auto ars = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)std::min<size_t>((size_t)M * lda * 2, 0x7fffffffu), 0x00020000);
  const auto brs = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)std::min<size_t>((size_t)M * ldb * 2, 0x7fffffffu), 0x00020000);
  const int nsteps = (m_end - m_begin + BM - 1) / BM;
  auto issue = [&](int s) {
    unsigned char* st = lds + (s % NS) * C::STAGE;
    const int m0 = m_begin + s * BM;
    const int valid = m_end - m0;               // >= BM except in a ragged last stage (uniform)
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const bool a = i < PA;
      unsigned vo = voff[i];
      if (valid < BM) {                         // rows past m_end re-read the last valid row
        const int row = (piece(i, a) * 1024 + lane * 16) / (a ? C::ROWA : C::ROWB);
        vo -= (unsigned)(row - min(row, valid - 1)) * (a ? lda : ldb) * 2u;
      }
      unsigned char* dst = st + (a ? 0 : C::A_BYTES) + piece(i, a) * 1024;
      if (a)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void*)dst, 16, vo,
                                                 (unsigned)m0 * lda * 2u, 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (__attribute__((address_space(3))) void*)dst, 16, vo,
                                                 (unsigned)m0 * ldb * 2u, 0, 0);
    }
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nsteps) issue(s);

  f32x16_t acc[RB][3];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // fused bias gradient: column sums of A (= dY) ride along on the k-tile-0 blocks: thread ->
  // (row tid / 16 (+16, ...), logical 16-byte chunk tid % 16), 8 running sums
  const bool do_bias = (dbias != nullptr) && bias_tile;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // transposed-read addressing: lane -> (half hh, column group cg, (q, pp));
  // rows of one read are 8 hh + q (+ 4 for the second), so the swizzle term is a per-lane constant
  const int hh = lane >> 5, cg = (lane >> 4) & 1, ii = lane & 15, q = ii >> 2, pp = ii & 3;
  const unsigned lds0 = (unsigned)(size_t)lds;
  unsigned a_addr[RB], b_addr[3];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int col = wn * 32 * RB + i * 32 + 16 * cg + 4 * pp;          // n column of the tile
    a_addr[i] = lds0 + (8 * hh + q) * C::ROWA + (((col >> 5) ^ C::swz_a(q)) << 6) + (col & 31) * 2;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int col = wk * 96 + j * 32 + 16 * cg + 4 * pp;               // k column of the tile
    // rows 8 hh + q and 8 hh + q + 4: (row >> 1) & 1 = (q >> 1) & 1 for both
    b_addr[j] = lds0 + C::A_BYTES + (8 * hh + q) * C::ROWB + (((col >> 5) ^ C::swz_b(q)) << 6) + (col & 31) * 2;
  }
  constexpr int KS = BM / 16;
  for (int s = 0; s < nsteps; ++s) {
    if (s + NS - 2 < nsteps - 0 && NS > 2 && s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // everyone's share of stage s has landed; everyone is done with stage s-1
    if (s + NS - 1 < nsteps) issue(s + NS - 1);
    const unsigned so = (unsigned)((s % NS) * C::STAGE);
    if (m_begin + (s + 1) * BM > m_end) {      // ragged last stage: the re-read rows count nothing
      const int valid = m_end - (m_begin + s * BM);
      for (int c = tid; c < (BM - valid) * (C::ROWA / 16); c += NT)
        *(uint4*)(lds + so + valid * C::ROWA + c * 16) = make_uint4(0, 0, 0, 0);
      __syncthreads();
    }
    if (do_bias) {
      constexpr int CPR = C::TN / 8, RP = NT / CPR;       // 16-byte chunks per A row; rows per pass of the workgroup
#pragma unroll
      for (int r0 = 0; r0 < BM; r0 += RP) {
        const int row = r0 + tid / CPR, lc = tid % CPR;
        uint4 u;    // (asm: a compiler-visible LDS load behind an LDS-DMA draws an s_waitcnt vmcnt(0))
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(u)
                     : "v"(lds0 + so + row * C::ROWA + ((((lc >> 2) ^ C::swz_a(row))) << 6) + (lc & 3) * 16) : "memory");
        bsum[0] += lo_bf16(u.x); bsum[1] += hi_bf16(u.x);
        bsum[2] += lo_bf16(u.y); bsum[3] += hi_bf16(u.y);
        bsum[4] += lo_bf16(u.z); bsum[5] += hi_bf16(u.z);
        bsum[6] += lo_bf16(u.w); bsum[7] += hi_bf16(u.w);
      }
    }
    // fragments of k-step ks: RB A fragments + 3 B fragments, two transposed reads each
    s16x4_t fl[2][RB + 3], fh[2][RB + 3];
    auto rd = [&](auto KS_, auto BUF_) {
      constexpr int ks = decltype(KS_)::value, buf = decltype(BUF_)::value;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        tn_read_tr<ks * 16 * C::ROWA>(fl[buf][i], a_addr[i] + so);
        tn_read_tr<(ks * 16 + 4) * C::ROWA>(fh[buf][i], a_addr[i] + so);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        tn_read_tr<ks * 16 * C::ROWB>(fl[buf][RB + j], b_addr[j] + so);
        tn_read_tr<(ks * 16 + 4) * C::ROWB>(fh[buf][RB + j], b_addr[j] + so);
      }
    };
    rd(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    tn_static_for<0, KS>([&](auto KS_) {
      constexpr int ks = decltype(KS_)::value, cur = ks & 1;
      if constexpr (ks + 1 < KS) {
        rd(std::integral_constant<int, ks + 1>{}, std::integral_constant<int, (cur ^ 1)>{});
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (RB + 3)) : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int f = 0; f < RB + 3; ++f) asm volatile("" : "+v"(fl[cur][f]), "+v"(fh[cur][f]));
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < RB; ++i)
          acc[i][j] = mfma32(make_bf16x8(fl[cur][i], fh[cur][i]), make_bf16x8(fl[cur][RB + j], fh[cur][RB + j]), acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = k0 + wk * 96 + j * 32 + (lane & 31);
      if (col >= K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = n0 + wn * 32 * RB + i * 32 + acc_row(r, lane);
#ifdef SVIT_TN_FLUSH_PLAIN     // (diagnostic builds, TIMING ONLY: what the fp32 atomic flush costs against plain stores)
        if (row < N) dW[(size_t)row * lddw + col] = acc[i][j][r];
#else
        if (row < N) atomicAdd(dW + (size_t)row * lddw + col, acc[i][j][r]);
#endif
      }
    }
  if (do_bias) {  // RP threads share a column chunk: reduce through LDS, one atomic per column
    constexpr int CPR = C::TN / 8, RP = NT / CPR;
    static_assert(BM % RP == 0, "whole passes over a stage");
    __syncthreads();                  // (the ring is read by nobody any more)
    float* red = (float*)lds;  // [RP][TN]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(tid / CPR) * C::TN + (tid % CPR) * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < C::TN && n0 + tid < N) {
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < RP; ++g) sum += red[g * C::TN + tid];
      atomicAdd(dbias + n0 + tid, sum);
    }
  }
#endif
}


constexpr int TN_BM = TnSmall::BM, TN_TN = TnSmall::TN, TN_TK = TnSmall::TK;   // the single-GEMM entry point

__global__ __launch_bounds__(256) void gemm_tn_kernel(const bf16_t* __restrict__ A, int lda,
                                                      const bf16_t* __restrict__ B, int ldb,
                                                      float* __restrict__ dW, int lddw, int M,
                                                      int N, int K, int rows_per_split,
                                                      float* __restrict__ dbias) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[TN_DMA_LDS];
  const int m_begin = blockIdx.z * rows_per_split;
  tn_tile_dma<TnSmallD>(lds, A, lda, B, ldb, dW, lddw, M, N, K, blockIdx.x * TN_TN, blockIdx.y * TN_TK, m_begin,
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
  int big[SVIT_TN_GROUP_MAX];          // 1: 128 x 192 tiles (TnBig), 0: 128 x 96 (TnSmall)
  int count;
};

#ifndef SVIT_TN_WPE         // waves per SIMD the grouped kernel is compiled for (diagnostic builds: 3 with SVIT_TN_BIG_NS=2 = three workgroups per CU)
#define SVIT_TN_WPE 2
#endif
__global__ __launch_bounds__(256, SVIT_TN_WPE) void gemm_tn_grouped_kernel(const TnGroup g) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[TN_DMA_LDS];
  // Logical ids are (problem, split)-major, tile-minor: the tiles of one split walk the SAME
  // rows in lock-step and re-read each other's A / B column panels (A once per k-tile, B once
  // per n-tile).  Consecutive logical ids are therefore placed on ONE XCD, so those re-reads
  // hit its L2 instead of crossing the fabric (rocprofv3 FETCH_SIZE before: 810 MB per launch
  // against ~230 MB of operands -- the kernel ran at HBM speed).
  const int nwg = gridDim.x, lin = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
  int pi = 0;
  const int bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
#pragma unroll
  for (int i = 1; i < SVIT_TN_GROUP_MAX; ++i)
    if (i < g.count && bid >= g.first_block[i]) pi = i;
  const int local = bid - g.first_block[pi];
  const int tile = local % g.tiles[pi], split = local / g.tiles[pi];
  const svit_tn_problem& p = g.p[pi];
  const int tn = tile % g.tiles_n[pi], tk = tile / g.tiles_n[pi];
  const int m_begin = split * g.rows_per_split[pi];
  if (g.big[pi])
    tn_tile_dma<TnBigD>(lds, (const bf16_t*)p.A, p.lda, (const bf16_t*)p.B, p.ldb, p.dW, p.lddw, p.M, p.N, p.K,
                        tn * TnBig::TN, tk * TnBig::TK, m_begin, min(p.M, m_begin + g.rows_per_split[pi]),
                        p.dbias, tk == 0);
  else
    tn_tile_dma<TnSmallD>(lds, (const bf16_t*)p.A, p.lda, (const bf16_t*)p.B, p.ldb, p.dW, p.lddw, p.M, p.N, p.K,
                          tn * TnSmall::TN, tk * TnSmall::TK, m_begin, min(p.M, m_begin + g.rows_per_split[pi]),
                          p.dbias, tk == 0);
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

// cost-model constants of the grouped launch: SVIT_K_TN_STEP_US_X100 (one 64-row step of a workgroup, 512 resident),
// SVIT_K_TN_ATOMIC_TBS_X100 (effective fp32 atomic flush rate) and the tile mode SVIT_K_TN_TILE of the knob table in
// common.h -- 2 = 128x192 everywhere since the in-step A/B of round 3 (13.46 -> 13.30 ms per step,
// profiles/r03_tn_tile_modes.txt: inside the step the operands come from HBM, not from the Infinity Cache an isolated
// loop keeps them in, and fewer, fatter tiles re-read less).  The 128x384 forms (eight waves, four waves, ring with loader
// waves) and the XCD-packed placement table of round 4 measured level or behind (profiles/r04_tn_l2_counters.txt) and
// live on only as tools/diag/variants/gemm_tn_r04_forms.hip.
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
    const int big_mode = svit_knob(SVIT_K_TN_TILE);
    const double step_us = svit_knob(SVIT_K_TN_STEP_US_X100) * 0.01, atomic_tbs = svit_knob(SVIT_K_TN_ATOMIC_TBS_X100) * 0.01;
    constexpr long slots = 512;      // resident 4-wave workgroups the planner counts on
    int bm[SVIT_TN_GROUP_MAX];
    double tile_bytes[SVIT_TN_GROUP_MAX];
    for (int i = 0; i < g.count; ++i) {
      g.p[i] = probs[base + i];
      // 128 x 192 tiles where they are fully used (K a multiple of 192) and measured faster with the
      // round-3 LDS-DMA body (tools/bench_kernels.py tngroup, profiles/r03_tn_tile_modes.txt): the
      // M = 3656 groups of blocks 14-15 (-5 / -11 %) and the M = 50696 groups of blocks 2-3 (-16 / -22 %);
      // not the M = 13064 groups of blocks 4-13 (+10 %: the doubled fp32 tile a workgroup flushes with
      // atomics outweighs the smaller operand traffic) nor the M = 201224 group of block 0 (+11 %)
      g.big[i] = big_mode == 1 ? (g.p[i].K % TnBig::TK == 0 && g.p[i].N >= 128 &&
                                  (g.p[i].M <= 4096 || (g.p[i].M >= 32768 && g.p[i].M < 131072)))
                 : big_mode == 3 ? (g.p[i].K % TnBig::TK == 0 && g.p[i].N >= 128)
                                 : (big_mode == 2);
      const int tn = g.big[i] ? TnBig::TN : TnSmall::TN, tk = g.big[i] ? TnBig::TK : TnSmall::TK;
      bm[i] = g.big[i] ? TnBigD::BM : TnSmall::BM;
      tile_bytes[i] = (double)tn * tk * 4.0;
      g.tiles_n[i] = (g.p[i].N + tn - 1) / tn;
      g.tiles[i] = g.tiles_n[i] * ((g.p[i].K + tk - 1) / tk);
      const long st = (g.p[i].M + bm[i] - 1) / bm[i];
      if (st > max_steps) max_steps = st;
    }
    // Every problem is cut into chunks of `steps` steps (64 reduction rows on 128 x 96 tiles, 32 on
    // 128 x 192: the same 12 MFMAs per wave), so all workgroups run about equally long.  Same
    // fitted model as svit_gemm_tn: 0.85 us per step with 512 resident workgroups, plus the atomic
    // flush of one fp32 tile per workgroup at ~0.75 TB/s.
    double best = 1e30;
    long best_steps = max_steps;
    for (long steps = 2; steps <= max_steps; steps += (steps < 32 ? 1 : steps / 16)) {
      long blocks = 0;
      double flush = 0.0;
      for (int i = 0; i < g.count; ++i) {
        const long st = (g.p[i].M + bm[i] - 1) / bm[i];
        const long nb = (long)g.tiles[i] * ((st + steps - 1) / steps);
        blocks += nb;
        flush += (double)nb * tile_bytes[i];
      }
      const double t = (double)((blocks + slots - 1) / slots) * steps * step_us +
                       flush / (atomic_tbs * 1e6);
      if (t < best) { best = t; best_steps = steps; }
    }
    if (ordered) best_steps = max_steps;
    int total = 0;
    for (int i = 0; i < g.count; ++i) {
      g.rows_per_split[i] = (int)best_steps * bm[i];
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
