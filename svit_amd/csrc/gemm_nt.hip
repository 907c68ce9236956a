// bf16 MFMA GEMM for the nn.Linear family, forward and dgrad -- gfx950.
//   svit_gemm_nt : C[M,N] = A[M,K] * W[N,K]^T with fused epilogues (bias / GELU / residual +
//                  DropPath / fp32 accumulate + row remap / GELU-backward), gemm_epilogue.h
//
// Main loop: multi-stage direct-to-LDS pipeline.  A/W tiles go HBM/L2 -> LDS with
// global_load_lds_dwordx4 (no VGPR staging, no ds_write), STAGES tiles of BK=32 are kept in
// flight behind a COUNTED s_waitcnt vmcnt(N) and ONE raw s_barrier per K-step, so the prefetch
// distance (STAGES-1 tiles) covers L2/HBM latency instead of the single tile a register-staged
// double buffer can hide (cdna guide: "Pipelining across barriers").  Every wave owns a
// (32*RB) x 96 output tile as 32x32x16 MFMA accumulators; tiles are walked in an XCD-aware
// order (N tiles of one A row panel stay on one XCD's L2).
//
// LDS image of a stage: rows of 64 B (32 bf16), lane-linear as global_load_lds requires
// (wave-uniform base + lane*16); the four 16-B chunks of a row are XOR-swizzled with
// ((row>>2)&3) on the per-lane SOURCE address, and the same XOR is applied on the fragment
// reads -> ds_read_b128 of 16 consecutive rows hits 16 distinct slots of the 256-B bank row.
#include "gemm_epilogue.h"
#include "attn_common.h"   // Int / static_for / lds_read128 / lgkm_release1

namespace {

// LDS byte offset of the 16-B chunk `ch` of tile row `r`.  BK = 32: rows of 64 B, chunk XOR
// ((r>>2)&3).  BK = 64: rows of 128 B (a whole 128-B cache line per row and K-step: a 64-B row
// uses half of every line it touches), two tile rows (r, r+8) share a 256-B bank row, chunk XOR
// (r&7) -> the 16 rows of a ds_read_b128 lane group land on 16 distinct 16-B bank slots.
template <int BK>
__device__ __forceinline__ int nt_lds_off(int r, int ch) {
  if constexpr (BK == 32) return r * 64 + 16 * (ch ^ ((r >> 2) & 3));
  else return (r >> 4) * 2048 + (r & 7) * 256 + ((r >> 3) & 1) * 128 + 16 * (ch ^ (r & 7));
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int RB, int NB, int WAVES_M, int WAVES_N, int STAGES, int EPI, int BK2>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, (WAVES_M * WAVES_N > 4 ? 1 : 2)) void
gemm_nt_v2_kernel(svit_gemm_args p) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = 32 * RB * WAVES_M, WN = 32 * NB, BN = WN * WAVES_N;
  // rows per stage, padded so that every wave issues the same number of loads per stage (the
  // counted vmcnt must be the same immediate for all waves); pad rows re-read a valid W row
  constexpr int CPR = BK2 / 8;                // 16-B chunks per tile row
  constexpr int RPR = NT / CPR;               // rows one round of loads (NT x 16 B) covers
  constexpr int ROWS = (BM + BN + RPR - 1) / RPR * RPR;
  constexpr int STAGE_BYTES = ROWS * BK2 * 2;
  constexpr int PER = ROWS * CPR / NT;        // global_load_lds per thread per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int nwg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
  const int wgid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
  const int m0 = (wgid / gridDim.x) * BM, n0 = (wgid % gridDim.x) * BN;

  // per-lane source pointers of this thread's PER chunks (k offset added per tile)
  const bf16_t* src[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int q = tid + i * NT;              // LDS position q*16 bytes within a stage
    int row, ch;
    if constexpr (BK2 == 32) {
      row = q >> 2;
      ch = (q & 3) ^ ((row >> 2) & 3);
    } else {
      const int line = q >> 4;
      row = (line >> 3) * 16 + ((q >> 3) & 1) * 8 + (line & 7);
      ch = (q & 7) ^ (row & 7);
    }
    if (row < BM) {
      const int gr = min(m0 + row, p.M - 1);
      src[i] = (const bf16_t*)p.A + (size_t)gr * p.lda + ch * 8;
    } else {
      const int gr = min(n0 + row - BM, p.N - 1);
      src[i] = (const bf16_t*)p.W + (size_t)gr * p.ldw + ch * 8;
    }
  }
  auto issue = [&](int kt, int stage) {
    unsigned char* base = smem + stage * STAGE_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < PER; ++i)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src[i] + kt * BK2),
          (__attribute__((address_space(3))) void*)(base + i * (NT * 16)), 16, 0, 0);
  };

  f32x16_t acc[RB][NB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK2;
  // prologue: STAGES-1 tiles in flight
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue(s, s);

  const int a_row = wm * 32 * RB + (lane & 31);
  const int w_row = BM + wn * WN + (lane & 31);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most the (STAGES-2) younger tiles' loads are outstanding
    if (kt + STAGES - 2 < nk) wait_vmcnt<(STAGES - 2) * PER>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // (a) everyone's part of tile kt is in LDS
                                    // (b) everyone finished reading tile kt-1's buffer
    if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
    const unsigned char* st = smem + (kt % STAGES) * STAGE_BYTES;
    // all fragment reads of the tile first, then the MFMAs back to back: left to itself the
    // compiler reused two fragment registers and put a full `s_waitcnt lgkmcnt(0)` in front of
    // nearly every MFMA (four exposed LDS round trips per tile)
    bf16x8_t af[BK2 / 16][RB], wf[BK2 / 16][NB];
#pragma unroll
    for (int ks = 0; ks < BK2 / 16; ++ks) {
      const int ch = 2 * ks + (lane >> 5);
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int r = a_row + i * 32;
        af[ks][i] = *(const bf16x8_t*)(st + nt_lds_off<BK2>(r, ch));
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int r = w_row + j * 32;
        wf[ks][j] = *(const bf16x8_t*)(st + nt_lds_off<BK2>(r, ch));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < BK2 / 16; ++ks)
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = mfma32(af[ks][i], wf[ks][j], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();   // all LDS reads of the last tiles done before the epilogue reuses LDS
  // (staging regions are per wave: no workgroup barrier inside the epilogue -- eight of them per 128x128 tile
  // kept the four waves in lock step through their stores)
  nt_epilogue<RB, NB, EPI, true>(p, acc, smem, m0, n0, wm, wn, lane, wave);
}

template <int RB, int NB, int WAVES_M, int WAVES_N, int STAGES, int BK2 = 32>
int launch_v2(const svit_gemm_args& a, hipStream_t st) {
  constexpr int BM = 32 * RB * WAVES_M, BN = 32 * NB * WAVES_N;
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int RPR = NT / (BK2 / 8);
  size_t lds = (size_t)STAGES * ((BM + BN + RPR - 1) / RPR * RPR) * BK2 * 2;
  const size_t lds_epi = (size_t)WAVES_M * WAVES_N * 16 * (32 * NB + 4) * sizeof(float);
  if (lds < lds_epi) lds = lds_epi;
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM);
  static SvitOnce once[6];
#define SVIT_V2_ATTR(E)                                                                              \
  if (int rc = svit_max_lds_once(once[E], (const void*)gemm_nt_v2_kernel<RB, NB, WAVES_M, WAVES_N, STAGES, E, BK2>, lds)) \
    return rc
  SVIT_V2_ATTR(SVIT_EPI_BF16); SVIT_V2_ATTR(SVIT_EPI_GELU); SVIT_V2_ATTR(SVIT_EPI_RESID);
  SVIT_V2_ATTR(SVIT_EPI_F32); SVIT_V2_ATTR(SVIT_EPI_DGELU); SVIT_V2_ATTR(SVIT_EPI_RELQ);
#undef SVIT_V2_ATTR
#define SVIT_V2_CASE(E)                                                                       \
  case E:                                                                                     \
    hipLaunchKernelGGL((gemm_nt_v2_kernel<RB, NB, WAVES_M, WAVES_N, STAGES, E, BK2>), grid, dim3(NT), \
                       lds, st, a);                                                           \
    break;
  switch (a.epilogue) {
    SVIT_V2_CASE(SVIT_EPI_BF16)
    SVIT_V2_CASE(SVIT_EPI_GELU)
    SVIT_V2_CASE(SVIT_EPI_RESID)
    SVIT_V2_CASE(SVIT_EPI_F32)
    SVIT_V2_CASE(SVIT_EPI_DGELU)
    SVIT_V2_CASE(SVIT_EPI_RELQ)
    default:
      return SVIT_ERR_ARG;
  }
#undef SVIT_V2_CASE
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

// ---- ring form: loader waves + MFMA waves (long-K GEMMs) ---------------------------------------------
// What bounds gemm_nt_v2_kernel on the long-K shapes is not a byte rate but the ISSUE of the LDS-DMA
// instructions: a 1-KiB piece holds the issuing wave for 60-185 cycles (MI355X guide, "LDS-DMA piece
// issue cost"), and in the v2 loop those cycles, the fragment reads and the MFMAs of a K-step are
// one wave's serial stream, so the matrix pipe idles while its wave feeds the ring (28 % busy, SQ counters
// of round 2) and only a second resident workgroup fills the holes.  Here the two jobs are different
// waves of one 8-wave workgroup, one of each per SIMD:
//   * waves 4..7 (loaders) do nothing but issue `global_load_lds_dwordx4` pieces (per-lane source pointers
//     fixed, one 64-bit add per piece and K-step) and wait for them with counted vmcnt; STAGES-1 K-steps
//     of 64 stay in flight;
//   * waves 0..3 (MFMA waves) own (32 RB) x (32 NB) accumulators and stream the fragments of one k16
//     sub-step AHEAD of the MFMAs that use them (inline-asm ds_read_b128, two per MFMA gap, released by
//     lgkmcnt(0) three MFMAs later), so neither the LDS latency nor any vector-memory instruction sits
//     in their instruction stream.
// One s_barrier per K-step joins them: B(kt) = "tile kt has landed" (loaders arrive after their vmcnt)
// and "tile kt-1 is read" (MFMA waves arrive after the lgkmcnt(0) of its last fragments) -- it sits
// BEFORE the last sub-step's MFMAs of tile kt-1, whose 32 RB NB cycles cover the first fragment reads of
// tile kt.  Tiles as in v2: BK = 64 rows of 128 B, chunk XOR (r & 7) (nt_lds_off<64>).
// The loader waves' whole life.  (Per-lane 64-bit source pointers + global_load_lds: the buffer-descriptor
// form of the same instruction, called with template-dependent operands from a kernel template, makes
// hipcc's HOST pass drop the kernel's stub without a diagnostic; the one v_lshl_add_u64 per piece this
// costs is nothing to a wave that does nothing else.)
template <int NL, int BM, int BN, int STAGES>
__device__ __forceinline__ void nt_ring_loader(const svit_gemm_args& p, unsigned char* smem, int lw, int lane,
                                               int m0, int n0, int nk) {
  constexpr int PIECES = (BM + BN) / 8;              // 1-KiB pieces of a stage image (8 rows x 128 B each)
  constexpr int PER = (PIECES + NL - 1) / NL;        // per loader wave and K-step (the same count for every
                                                     // wave: a surplus slot re-loads the last piece)
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  // piece q = lane-linear 16-B slots; slot -> (tile row, 16-B chunk) is the inverse of nt_lds_off<64>;
  // a piece lies wholly in A rows or wholly in W rows (BM % 16 == 0)
  const bf16_t* src[PER];
  auto piece_of = [&](int i) { return min(lw + NL * i, PIECES - 1); };
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int q = piece_of(i) * 64 + lane, line = q >> 4;
    const int row = (line >> 3) * 16 + ((q >> 3) & 1) * 8 + (line & 7);
    const int ch = (q & 7) ^ (row & 7);
    if (row < BM) src[i] = (const bf16_t*)p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + ch * 8;
    else src[i] = (const bf16_t*)p.W + (size_t)min(n0 + row - BM, p.N - 1) * p.ldw + ch * 8;
  }
  auto issue = [&](int kt) {
    unsigned char* st = smem + (kt % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < PER; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + kt * 64),
                                       (__attribute__((address_space(3))) void*)(st + piece_of(i) * 1024), 16, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < STAGES; ++s)
    if (s < nk) issue(s);
  // B(kt): tile kt has landed once at most the tiles issued after it are outstanding
  for (int kt = 0; kt < nk; ++kt) {
    // tiles issued after tile kt so far: the prologue's, then one per passed barrier
    const int younger = (kt == 0 ? min(STAGES - 1, nk - 1) : min(kt + STAGES - 2, nk - 1) - kt);
    attn::static_for<0, STAGES>([&](auto Y) {
      if (younger == decltype(Y)::value) wait_vmcnt<decltype(Y)::value * PER>();
    });
    __builtin_amdgcn_s_barrier();
    // the barrier also says tile kt-1 is read: its slot takes tile kt-1+STAGES
    if (kt >= 1 && kt - 1 + STAGES < nk) issue(kt - 1 + STAGES);
  }
}

template <int RB, int NB, int WAVES_M, int WAVES_N, int NL, int STAGES, int EPI>
__global__ __launch_bounds__((WAVES_M * WAVES_N + NL) * 64, 1) void gemm_nt_ring_kernel(svit_gemm_args p) {
  using attn::Int;
  constexpr int NC = WAVES_M * WAVES_N;          // MFMA waves; NL loader waves behind them
  constexpr int BM = 32 * RB * WAVES_M, WN = 32 * NB, BN = WN * WAVES_N;
  static_assert(BM % 16 == 0 && BN % 16 == 0, "a 1-KiB piece must not straddle A and W rows");
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
  const int wgid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
  const int m0 = (wgid / gridDim.x) * BM, n0 = (wgid % gridDim.x) * BN;
  const int nk = p.K / 64;

  if (wave >= NC) {
    nt_ring_loader<NL, BM, BN, STAGES>(p, smem, wave - NC, lane, m0, n0, nk);
    return;
  }

  // -------------------------------------------------- MFMA waves ----------------------------------------
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  f32x16_t acc[RB][NB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: row r = 16 g + (lane & 31) (+ 32 per row block), chunk 2 ks + (lane >> 5);
  // chunk ^ (r & 7) = 2 (ks ^ ((r & 7) >> 1)) + ((lane >> 5) ^ (r & 1)): one address per ks, row blocks and
  // the A / W split are immediates or wave constants
  const int l31 = lane & 31, x7 = lane & 7;
  unsigned fa[4], fw[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const unsigned lp = (unsigned)((l31 >> 4) * 2048 + x7 * 256 + ((l31 >> 3) & 1) * 128 +
                                   32 * (ks ^ (x7 >> 1)) + 16 * ((lane >> 5) ^ (x7 & 1)));
    fa[ks] = (unsigned)(size_t)smem + lp + (unsigned)(wm * 2 * RB) * 2048u;
    fw[ks] = (unsigned)(size_t)smem + lp + (unsigned)((BM + wn * WN) / 16) * 2048u;
  }
  bf16x8_t fr[2][RB + NB];       // fragments of two consecutive k16 sub-steps (A row blocks, then W)
  auto reads = [&](auto KS, auto BUF, unsigned so, auto HALF) {
    // HALF 0: the first (RB+NB+1)/2 fragments, HALF 1: the rest -- two reads per MFMA gap
    constexpr int ks = decltype(KS)::value, b = decltype(BUF)::value, h = decltype(HALF)::value;
    attn::static_for<0, RB + NB>([&](auto F) {
      constexpr int f = decltype(F)::value;
      if constexpr ((f / 2) == h) {
        if constexpr (f < RB) attn::lds_read128<f * 4096>(fr[b][f], fa[ks] + so);
        else attn::lds_read128<(f - RB) * 4096>(fr[b][f], fw[ks] + so);
      }
    });
  };
  constexpr int NH = (RB + NB + 1) / 2;       // read groups of two
  auto release = [&](auto BUF) {
    constexpr int b = decltype(BUF)::value;
    attn::static_for<0, RB + NB>([&](auto F) { attn::lgkm_release1<0>(fr[b][decltype(F)::value]); });
  };
  // MFMAs of one sub-step from buffer B, with the reads of the next sub-step (buffer 1-B) in the first gaps
  auto substep = [&](auto BUF, auto NEXT_KS, unsigned next_so, bool do_reads) {
    constexpr int b = decltype(BUF)::value;
    attn::static_for<0, RB * NB>([&](auto Q) {
      constexpr int q = decltype(Q)::value, i = q / NB, j = q % NB;
      acc[i][j] = mfma32(fr[b][i], fr[b][RB + j], acc[i][j]);
      if constexpr (q < NH) {
        if (do_reads) reads(NEXT_KS, Int<1 - b>{}, next_so, Int<q>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  __builtin_amdgcn_s_barrier();                       // B(0)
  attn::static_for<0, NH>([&](auto H) { reads(Int<0>{}, Int<0>{}, 0u, H); });
  release(Int<0>{});
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned so = (unsigned)((kt % STAGES) * STAGE_BYTES);
    const unsigned so_next = (unsigned)(((kt + 1) % STAGES) * STAGE_BYTES);
    const bool more = kt + 1 < nk;
    substep(Int<0>{}, Int<1>{}, so, true);
    release(Int<1>{});
    substep(Int<1>{}, Int<2>{}, so, true);
    release(Int<0>{});
    substep(Int<0>{}, Int<3>{}, so, true);
    release(Int<1>{});                                // every read of tile kt has returned
    if (more) __builtin_amdgcn_s_barrier();           // B(kt+1): tile kt+1 is in LDS, tile kt's slot is free
    substep(Int<1>{}, Int<0>{}, so_next, more);
    if (more) release(Int<0>{});
  }
  // staging for the epilogue: a slot no MFMA wave can still be reading (tile nk-2's; all DMA has landed)
  unsigned char* epi = smem + (nk >= 2 ? (nk - 2) % STAGES : 1) * STAGE_BYTES;
  nt_epilogue<RB, NB, EPI, true>(p, acc, epi, m0, n0, wm, wn, lane, wave);
}

template <int RB, int NB, int WAVES_M, int WAVES_N, int NL, int STAGES>
int launch_ring(const svit_gemm_args& a, hipStream_t st) {
  constexpr int BM = 32 * RB * WAVES_M, BN = 32 * NB * WAVES_N, NT = (WAVES_M * WAVES_N + NL) * 64;
  constexpr size_t lds = (size_t)STAGES * (BM + BN) * 128;
  static_assert((size_t)WAVES_M * WAVES_N * 16 * (32 * NB + 4) * sizeof(float) <= (size_t)(BM + BN) * 128, "epilogue staging fits a slot");
  static_assert(lds <= 160 * 1024, "LDS");
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM);
  static SvitOnce once[5];
#define SVIT_RING_ATTR(E)                                                                              \
  if (int rc = svit_max_lds_once(once[E], (const void*)gemm_nt_ring_kernel<RB, NB, WAVES_M, WAVES_N, NL, STAGES, E>, lds)) \
    return rc
  SVIT_RING_ATTR(SVIT_EPI_BF16); SVIT_RING_ATTR(SVIT_EPI_GELU); SVIT_RING_ATTR(SVIT_EPI_RESID);
  SVIT_RING_ATTR(SVIT_EPI_F32); SVIT_RING_ATTR(SVIT_EPI_DGELU);
#undef SVIT_RING_ATTR
#define SVIT_RING_CASE(E)                                                                       \
  case E:                                                                                       \
    hipLaunchKernelGGL((gemm_nt_ring_kernel<RB, NB, WAVES_M, WAVES_N, NL, STAGES, E>), grid, dim3(NT), lds, st, a); \
    break;
  switch (a.epilogue) {
    SVIT_RING_CASE(SVIT_EPI_BF16)
    SVIT_RING_CASE(SVIT_EPI_GELU)
    SVIT_RING_CASE(SVIT_EPI_RESID)
    SVIT_RING_CASE(SVIT_EPI_F32)
    SVIT_RING_CASE(SVIT_EPI_DGELU)
    default:
      return SVIT_ERR_ARG;
  }
#undef SVIT_RING_CASE
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}
}  // namespace

// tuning knobs: SVIT_K_NT_STAGES / _CFG / _BK of the table in common.h (svit_debug_set, tools and variant-parity tests only)
// Which ring kernel, if any.  Two sets of measurements on MI355X decide (profiles/r03_nt_ring.txt):
// isolated loops per epilogue family (tools/bench_kernels.py ntring <epilogue>), where the ring kernels win
// almost everywhere K % 64 == 0, and the replayed training step with every NT launch matched by position
// (tools/nt_by_position.py), where the short-K wins do NOT survive (a launch's neighbours are other kernels,
// not copies of itself: 128x128 / 2 stages for fc1 was +5 % there, 128x96 / 2 stages at K = 384 +11 %).
// Kept are the two rules that win in both: *cfg stays < 0 otherwise (v2 kernels).
static void ring_choice(const svit_gemm_args& a, int* cfg, int* stages) {
  const long tm = (a.M + 127) / 128;
  const long t192 = a.N % 192 == 0 ? tm * (a.N / 192) : 0, t96 = tm * ((a.N + 95) / 96);
  // few, long tiles: every 128x96 tile has a CU to itself -> 4-deep ring (3656 x 768 x 768..3072: -30 %)
  constexpr int k_min5 = 1024, t_max6 = 256, s6 = 4;     // (swept inside the step in round 4: profiles/r04_in_step_sweeps.txt)
  if (t96 <= t_max6 && a.K >= 768) { *cfg = 6, *stages = (t96 > 256 ? 2 : s6); return; }   // (> 256 tiles: two workgroups per CU need the 2-stage ring)
  // long K, narrow output, enough 128x192 tiles for most CUs (13064 x 384 x 1152..2304: -6..-10 % in the
  // step, -15..-20 % isolated; on a par with hipBLASLt's 128x160x64 macro-tile)
  if (t192 >= 160 && a.N <= 768 && a.K >= k_min5) { *cfg = 5, *stages = 3; return; }
}

extern "C" int svit_gemm_nt(const svit_gemm_args* args, void* stream) {
  if (!args || !args->A || !args->W) return SVIT_ERR_ARG;
  const svit_gemm_args& a = *args;
  const bool relq = a.epilogue == SVIT_EPI_RELQ;
  if (!relq && !a.out) return SVIT_ERR_ARG;
  if (relq) {
    if (!a.relq_map || !a.relq_out || (a.relq_extra != 32 && a.relq_extra != 64) || a.relq_rows < 1 ||
        a.relq_ld < 96 + a.relq_extra || a.bias)
      return SVIT_ERR_ARG;
  }
  if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.K % 32 != 0 || a.N % 96 != 0) return SVIT_ERR_SHAPE;
  if (a.lda % 8 != 0 || a.ldw % 8 != 0 || a.lda < a.K || a.ldw < a.K || (!relq && a.ldo < a.N))
    return SVIT_ERR_ALIGN;
  if (((uintptr_t)a.A | (uintptr_t)a.W) & 15) return SVIT_ERR_ALIGN;
  if (!relq && (a.ldo % 4 != 0 || ((uintptr_t)a.out & 15))) return SVIT_ERR_ALIGN;
  if (a.bias && ((uintptr_t)a.bias & 15)) return SVIT_ERR_ALIGN;
  if (a.aux && (a.ldaux % 4 != 0 || ((uintptr_t)a.aux & 15))) return SVIT_ERR_ALIGN;
  if (a.out2 && (a.ldo2 % 4 != 0 || ((uintptr_t)a.out2 & 15))) return SVIT_ERR_ALIGN;
  if ((a.epilogue == SVIT_EPI_RESID || a.epilogue == SVIT_EPI_DGELU) && !a.aux) return SVIT_ERR_ARG;
  if (a.epilogue == SVIT_EPI_RESID && a.row_scale && a.rows_per_sample <= 0) return SVIT_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  // Tile choice (measured on MI355X, tools/bench_kernels.py ntstages): 128x192 blocks (2x2 waves
  // of 64x96) win whenever they still give >= 1 tile per CU; otherwise, and for N = 96 (mod
  // 192), 128x96 blocks (4 waves of 32x96) double the number of workgroups.  A 256x192 / 8-wave
  // tile (fewer LDS-fill bytes per flop) was measured too and never won.  At M = 13064 every
  // variant lands within 10 % of the others: the bound is the per-CU LDS fill rate (~40 GB/s
  // per CU, ~10 TB/s chip-wide for the mix of L2 and Infinity-Cache hits), not the pipeline
  // depth, the fragment-read scheduling or the tile shape.  Also measured and rejected: a
  // persistent form (a workgroup walks several tiles and issues the next tile's first LDS-DMA from
  // inside the epilogue of the current one; commit "experiment: persistent NT GEMM", profiles/
  // r02_nt_persistent_tiles.txt): -10..-17 % on the many-tile K = 384 shapes of the 56x56 stage,
  // +5..10 % wherever the epilogue writes two outputs or fp32 rows -- the next tile's first
  // s_waitcnt vmcnt(0) also waits for this tile's stores, which a finishing workgroup never does.
  // Net zero over the step; not kept.
  // register-staged form (global_load_dwordx4 -> VGPR -> ds_write_b128, K-step 64, loads a whole
  // K-step ahead; commit "experiment: register-staged NT GEMM variant", profiles/
  // r02_nt_tile_sweep.txt column REG): 5-10 % slower than LDS-DMA on every shape -- the bound is
  // the CU's vector-memory path itself (SQ counters, profiles/r02_nt_pmc_13064x384x1536.txt: waves
  // spend 51 % of their cycles stalled at instruction issue with the MFMA pipe 28 % busy; ~35
  // cycles per 1-KiB load instruction per CU = 29 B/clk/CU = 56 GB/s/CU), not how the bytes reach
  // LDS.  Also measured and rejected: a row-stationary form for K <= 384 (8-wave workgroups, A rows held in registers as MFMA
  // fragments, only W streamed through a 4-deep LDS ring: 4x fewer fill bytes per flop) --
  // 0.6-0.9x the speed of these tiles on every shape of the model: one barrier-locked workgroup
  // per CU leaves nothing to overlap its epilogues and barriers with.
  // force_cfg: 0 / 2 / 4 = v2 tiles, 5 / 6 / 7 = ring tiles, 8 = the v2 heuristic (no ring), -1 = heuristic
  const int force_raw = svit_knob(SVIT_K_NT_CFG), force_stages = svit_knob(SVIT_K_NT_STAGES);
  const bool no_ring = force_raw == 8 || svit_knob(SVIT_K_NT_BK) != 0;
  const int force_cfg = force_raw == 8 ? -1 : force_raw;
  // Ring kernels (loader waves + MFMA waves, K-steps of 64; tools/bench_kernels.py ntring, profiles/
  // r03_nt_ring.txt): see ring_choice().
  if (a.K % 64 == 0 && !relq && ((force_cfg < 0 && !no_ring) || (force_cfg >= 5 && force_cfg <= 7))) {
    int cfg = force_cfg, rs = force_stages;
    if (cfg < 0) ring_choice(a, &cfg, &rs);
    if (cfg >= 5) {
      if (!rs) rs = 3;
#define SVIT_RING_PICK(RB, NB, WM, WN)                                         \
  do {                                                                         \
    if (rs == 2) return launch_ring<RB, NB, WM, WN, 4, 2>(a, st);              \
    if (rs == 3) return launch_ring<RB, NB, WM, WN, 4, 3>(a, st);              \
    return launch_ring<RB, NB, WM, WN, 4, 4>(a, st);                           \
  } while (0)
      if (cfg == 5) SVIT_RING_PICK(2, 3, 2, 2);
      if (cfg == 6) SVIT_RING_PICK(1, 3, 4, 1);
      SVIT_RING_PICK(2, 2, 2, 2);
#undef SVIT_RING_PICK
    }
  }
  bool big = a.N % 192 == 0 && (long)((a.M + 127) / 128) * (a.N / 192) >= 256;
  if (force_cfg == 0 && a.N % 192 == 0) big = true;
  if (force_cfg == 2) big = false;
  // pipeline depth (measured, tools/bench_kernels.py ntstages): short K loops prefer 2 stages
  // (less LDS -> more resident workgroups), K >= 2048 wants the 4-deep prefetch
  const int stages = force_stages ? force_stages : (a.K >= 2048 ? 4 : 2);
  // 128x128 blocks (2x2 waves of 64x64): measured ~10 % faster than 128x192 for the wide, short-K
  // GEMMs of the 14x14 stage (M = 13064, N = 1152 / 1536, K = 384: fc1, fc2-dgrad, qkv)
  bool sq = a.N % 128 == 0 && a.N >= 1024 && a.K <= 512 &&
            (long)((a.M + 127) / 128) * (a.N / 128) <= 2048;
  if (force_cfg == 4) sq = a.N % 128 == 0;
  else if (force_cfg >= 0) sq = false;
  // K-step of 64 (whole 128-B lines per tile row; half as many barriers): measured 5-14 % faster
  // for the narrow-output, long-K GEMMs (fc2, proj, qkv dgrad: N <= 768, K >= 384) as 128x96
  // blocks at depth 2, and slower elsewhere -- the 2x LDS per stage costs a resident workgroup
  // (tools/bench_kernels.py ntstages, profiles/r02_nt_tile_sweep.txt)
  const int force_bk = svit_knob(SVIT_K_NT_BK);
  bool bk64 = a.K % 64 == 0 && a.N <= 768 && a.K >= 384;
  if (force_bk) bk64 = force_bk == 64 && a.K % 64 == 0;
  if (bk64 && force_cfg < 0 && !force_stages) return launch_v2<1, 3, 4, 1, 2, 64>(a, st);
#define SVIT_NT_PICK(RB, NB, WM, WN)                                           \
  do {                                                                         \
    if (bk64) {                                                                \
      if (stages == 2) return launch_v2<RB, NB, WM, WN, 2, 64>(a, st);         \
      if (stages == 3) return launch_v2<RB, NB, WM, WN, 3, 64>(a, st);         \
      return launch_v2<RB, NB, WM, WN, 4, 64>(a, st);                          \
    }                                                                          \
    if (stages == 2) return launch_v2<RB, NB, WM, WN, 2, 32>(a, st);           \
    if (stages == 3) return launch_v2<RB, NB, WM, WN, 3, 32>(a, st);           \
    return launch_v2<RB, NB, WM, WN, 4, 32>(a, st);                            \
  } while (0)
  // round 4: ONE-ROUND tiles for the wide short-K GEMMs of the 14x14 stage.  At M = 13064 the 128-row tiles above put
  // 824 (128x192) or 1236 (128x128) workgroups on 512-768 slots: a second, part-filled round in which the chip mostly
  // stores.  160x256 (1 x 4 waves of 160x64) covers 13064 x 1536 with 82 x 6 = 492 workgroups, 192x192 (2 x 2 waves of
  // 96x96) covers 13064 x 1152 with 69 x 6 = 414: every workgroup resident at once (two per CU), more flops per LDS-fill
  // byte (91 / 96 against 77 / 64).  force_cfg 9 / 10 force them; heuristic: only where the grid is one
  // round and the 128-row grid is not.
  {
    constexpr int one_round = 1;
    const long t160 = a.N % 256 == 0 ? (long)((a.M + 159) / 160) * (a.N / 256) : 0;
    const long t192 = a.N % 192 == 0 ? (long)((a.M + 191) / 192) * (a.N / 192) : 0;
    const long t128 = (long)((a.M + 127) / 128) * ((a.N + 191) / 192);
    const bool auto_ok = force_cfg < 0 && one_round && !force_stages && !force_bk && a.K <= 512 && t128 > 512;
    if (force_cfg == 9 || (auto_ok && t160 > 256 && t160 <= 512)) {
      if (a.N % 256 == 0) return launch_v2<5, 2, 1, 4, 2, 32>(a, st);
    }
    if (force_cfg == 10 || (auto_ok && t192 > 256 && t192 <= 512)) {
      if (a.N % 192 == 0) return launch_v2<3, 3, 2, 2, 2, 32>(a, st);
    }
  }
  if (sq) SVIT_NT_PICK(2, 2, 2, 2);
  if (big) SVIT_NT_PICK(2, 3, 2, 2);
  SVIT_NT_PICK(1, 3, 4, 1);
#undef SVIT_NT_PICK
}
