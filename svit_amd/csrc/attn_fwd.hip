// Fused pooled attention, forward (attention.py:429-461) -- gfx950.
//
//   S = (q*scale) k^T + rel-pos bias ; P = softmax(S) ; ctx = P v + q (all tokens but cls)
//
// The decomposed relative-position bias rides inside the QK^T MFMA, and the product comes out in
// the log2 domain: the query operand is augmented to qa = [q | log2e * relq] and the key operand to
// ka = [scale * log2e * k | one-hot(y,x,t)] (the pooling kernel writes the keys pre-multiplied), so
// qa.ka^T = log2e * (scale*q.k + rel_h[y] + rel_w[x] + rel_t[t]); cls/object rows/cols carry zeros
// in the bias columns.  Head dim of the contraction DA = 128 or 160, value dim 96.
//
// Work split: block = 4 waves x 32 queries; K/V tiles of 64 keys arrive by LDS-DMA
// (global_load_lds) in a two-stage, swizzled LDS panel image (attn_common.h), one raw barrier
// per tile behind the wave's own vmcnt wait.  Scores are
// computed "swapped" (S^T = ka qa^T): every lane owns ONE query column and 16 key rows per
// 32x32 block, so the online-softmax row reduction is in-register plus one lane<->lane+32
// exchange, and the exponentiated tile feeds the PV MFMA as its B operand without leaving
// registers.  O^T accumulates as 3 x (32 dv x 32 query) blocks.
#include <algorithm>
#include "attn_common.h"
#include "../../include/svit_hip.h"

#ifdef SVIT_ATTN_STAMPS   // tools/attn_stamps.py: cycle stamps of workgroup 0 / wave 0 per tile phase
__device__ unsigned long long g_attn_stamps[8192];
extern "C" int svit_debug_attn_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_stamps), sizeof(unsigned long long) * n);
}
#define STAMP(slot)                                                                     \
  do {                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    if (stamp_on) g_attn_stamps[(slot)] = __builtin_readcyclecounter();                 \
    __builtin_amdgcn_sched_barrier(0);                                                  \
  } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

// Diagnostic builds only (tools/diag/build_variant.py ... -DSVIT_ATTN_ABL=<mask>): anatomy of the per-launch fixed
// cost by ablation -- 1: no Q row loads, 2: no residual-row loads, 4: no ctx stores, 8: no K/V DMA (and no waits).
// Results are wrong by construction; never defined in the product build.
#ifndef SVIT_ATTN_ABL
#define SVIT_ATTN_ABL 0
#endif
#ifndef SVIT_ATTN_QSTAGE      // 0 in a diagnostic build: the round-3 row-per-lane Q loads (A/B)
#define SVIT_ATTN_QSTAGE 1
#endif

namespace {
using namespace attn;
constexpr int KT = 64;  // keys per tile

__device__ __forceinline__ float other_half(float x) {      // lane l <-> lane l ^ 32, VALU only
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

// NW waves per workgroup share one K/V stream (NW*32 queries); NS = depth of the K/V ring.
// <4, 2>: two independent workgroups per CU.  <8, 3>: one 8-wave workgroup per CU streams each
// K/V tile ONCE for 256 queries and prefetches two tiles ahead.
// KSU = k-steps of the QK^T contraction that carry data: 6 (q.k) + ceil(bias columns / 16); the
// K image holds ceil(KSU / 2) panels.
// Round 3 (the loop's VALU work went from ~110 to ~75 instructions per tile and wave):
//   * the score leaves the MFMA in the log2 domain (the pooling kernel writes the keys multiplied
//     by scale * log2 e) and already relative to the running maximum (-m enters as the initial
//     accumulator of the QK^T chain: a block of 16 registers rewritten only when the maximum is
//     re-based), so a probability costs ONE v_exp_f32 -- no multiply, no subtract;
//   * every fragment read goes through the RowStream / TrStream ring (attn_common.h) four / three
//     fragments ahead of its MFMA (was: chunks of 4-7 fragments, twice the registers);
//   * a ragged last tile with at most 32 valid keys multiplies half a tile (Nk = 457: 7.5 tiles
//     instead of 8), and is fetched through clamped addresses (nothing relies on what an LDS-DMA
//     does past the end of a buffer);
//   * the residual-pooling rows are fetched after the loop (24 registers less inside it).
// Kept from round 2: row sums of P on the matrix pipe (one extra MFMA per 16 keys against a
// constant "ones" fragment), LDS-DMA pieces addressed by buffer descriptor + scalar offset, only
// the k-steps with data multiplied, output through LDS as whole 192-byte rows with 16-byte stores.
template <int KSU, int NW, int NS>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void attn_fwd_kernel(svit_attn_fwd_args a) {
#if __HIP_DEVICE_COMPILE__          // (the buffer-descriptor builtins have no host-side type)
  constexpr int NP = (KSU + 1) / 2, KCOLS = NP * 32;
  constexpr int K_BYTES = KT * KCOLS * 2, V_BYTES = KT * HD * 2;
  constexpr int STAGE = K_BYTES + V_BYTES;    // [K tile | V tile] per pipeline stage
  using KLoad = BufTile<KT, KCOLS, NW>;
  using VLoad = BufTile<KT, HD, NW>;
  constexpr int PIECES = KLoad::PER_WAVE + VLoad::PER_WAVE;   // DMA instructions per wave and tile
  using QStage = RowStage<(KSU <= 8 ? 128 : 160)>;
  constexpr bool QSTAGE = (SVIT_ATTN_ABL & 1) == 0 && SVIT_ATTN_QSTAGE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
#ifdef SVIT_ATTN_STAMPS
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) g_attn_stamps[4] = __builtin_readcyclecounter();
#endif
  const int DA = a.DA;
  const int wgid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int bh = wgid / gridDim.x, b = bh / a.heads, head = bh % a.heads;
  const int q0 = (wgid % gridDim.x) * (NW * 32) + wave * 32;
  const int qi = q0 + (lane & 31);
  const int qc = min(qi, a.Nq - 1);
  const bf16_t* qa = (const bf16_t*)a.qa + ((size_t)bh * a.Nq) * DA;
  const bf16_t* ka = (const bf16_t*)a.ka + ((size_t)bh * a.Nk) * DA;
  const bf16_t* vv = (const bf16_t*)a.v + ((size_t)bh * a.Nk) * HD;

  f32x16_t o[3], lacc, negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) { lacc[r] = 0.f; negm[r] = 0.f; }
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
  // A operand of the row-sum MFMA: output row 0 = sum over the 16 keys of the P fragment
  bf16x8_t onesf;
#pragma unroll
  for (int e = 0; e < 8; ++e) onesf[e] = (lane & 31) == 0 ? (__bf16)1.0f : (__bf16)0.0f;
  float m_run = 0.f;            // running max (log2 domain); negm holds -m_run in all 16 registers

  // per-lane LDS byte addresses of the fragment reads (stage 0; everything else is an
  // immediate): K row fragments of k-step ks sit at kaddr[ks&1] + (ks>>1)*KT*64 + kb*2048, the
  // transposed V fragments of key group rbase / panel j at vaddr[0|1] + rbase*64 + j*KT*64
  // (image and swizzle: attn_common.h).
  const unsigned lds0 = (unsigned)(size_t)smem;
  unsigned kaddr0[2], vaddr0[2];
  {
    const int row = lane & 31, sw = (row >> 2) & 3;
    kaddr0[0] = lds0 + row * 64 + 16 * ((0 + hh) ^ sw);
    kaddr0[1] = lds0 + row * 64 + 16 * ((2 + hh) ^ sw);
    const int cg = (lane >> 4) & 1, i = lane & 15, q = i >> 2, pp = i & 3;
    const int r0 = 4 * hh, ch = 2 * cg + (pp >> 1);
    const unsigned vb = lds0 + K_BYTES + 8 * (pp & 1);
    vaddr0[0] = vb + (r0 + q) * 64 + 16 * (ch ^ ((r0 >> 2) & 3));
    vaddr0[1] = vb + (r0 + 8 + q) * 64 + 16 * (ch ^ (((r0 + 8) >> 2) & 3));
  }

#ifdef SVIT_ATTN_STAMPS
  const bool stamp_on = blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
  if (stamp_on) { g_attn_stamps[0] = __builtin_readcyclecounter(); g_attn_stamps[1] = wall_clock64(); }
#endif
  const int nt = (a.Nk + KT - 1) / KT;
  KLoad kload;
  VLoad vload;
  kload.init(DA, wave, lane);
  vload.init(HD, wave, lane);
  // descriptors over this (batch, head)'s K and V; the ragged last tile re-reads the last valid
  // row through clamped per-lane offsets (those keys are masked to -inf below)
  const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)ka, 0, a.Nk * DA * 2, 0x00020000);
  const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vv, 0, a.Nk * HD * 2, 0x00020000);
  auto issue = [&](int t) {
    if constexpr ((SVIT_ATTN_ABL & 8) != 0) return;
    unsigned char* st = smem + (t % NS) * STAGE;
    const unsigned k0 = (unsigned)t * KT;
    kload.issue_auto(krs, k0 * DA * 2u, DA, a.Nk - (int)k0, st, wave, lane);
    vload.issue_auto(vrs, k0 * HD * 2u, HD, a.Nk - (int)k0, st + K_BYTES, wave, lane);
  };
  issue(0);                        // the first tile(s) travel while the Q fragments are fetched
  if (NS == 3 && nt > 1) issue(1);
  bf16x8_t qf[KSU];
  if constexpr (QSTAGE) {
    // round 4: the wave's 32 Q rows come through LDS, coalesced (attn_common.h RowStage), into a region of their
    // own behind the stages the prologue fills (NS - 1 of them) -- the last stage and the tail of the allocation,
    // free until the next tile is issued after the first barrier, which every wave reaches only after it has read
    // its fragments
    unsigned char* qreg = smem + (NS - 1) * STAGE + wave * QStage::BYTES;
    QStage::issue(qa, DA, q0, a.Nq, qreg, lane);
    wait_vmcnt<0>();
#pragma unroll
    for (int ks = 0; ks < KSU; ++ks) qf[ks] = QStage::frag(qreg, ks, lane);
  } else {
#pragma unroll
    for (int ks = 0; ks < KSU; ++ks) {
      if constexpr ((SVIT_ATTN_ABL & 1) != 0) {
        for (int e = 0; e < 8; ++e) qf[ks][e] = (__bf16)(0.01f * (float)(lane + ks));
      } else {
        qf[ks] = *(const bf16x8_t*)(qa + (size_t)qc * DA + ks * 16 + hh * 8);
      }
    }
  }
  // pin the register operands before the tile loop: their first use must not sit inside it,
  // or the compiler's wait for them (vmcnt(0)) would drain the LDS-DMA pipeline every tile
  // (here that wait also covers the tiles issued above, which tile 0 needs anyway)
#pragma unroll
  for (int ks = 0; ks < KSU; ++ks) asm volatile("" : "+v"(qf[ks]));
  static_prio(blockIdx.y * gridDim.x + blockIdx.x, wave, NW);

  // One K/V tile.  HALF: a ragged last tile with at most 32 valid keys -- only key block 0 is
  // multiplied (QK^T, softmax and P.V of 32 keys).
  auto tile = [&](int t, auto HalfTag) {
    constexpr bool HALF = decltype(HalfTag)::value == 1;
    constexpr int NKB = HALF ? 1 : 2;
    // this wave's share of tile t has landed (NS == 3: tile t+1's pieces may stay in flight)
    if (NS == 3 && t + 1 < nt) wait_vmcnt<PIECES>();
    else wait_vmcnt<0>();
    STAMP(8 + t * 8 + 5);
    __builtin_amdgcn_s_barrier();    // everyone's share has; everyone is done with tile t-1
    STAMP(8 + t * 8 + 0);
    if (t + NS - 1 < nt) issue(t + NS - 1);    // into tile t-1's slot; travels while tiles are consumed
    STAMP(8 + t * 8 + 1);
    const unsigned so = (t % NS) * STAGE;
    const unsigned kaddr[2] = {kaddr0[0] + so, kaddr0[1] + so};
    const unsigned vaddr[2] = {vaddr0[0] + so, vaddr0[1] + so};
    // V^T fragments: item i = 3 * g + j is key group g (16 keys), value panel j
    TrStream<6 * NKB, 3> vs;
    auto rdv = [&](auto J, s16x4_t& lo, s16x4_t& hi) {
      constexpr int i = decltype(J)::value, g = i / 3, j = i % 3;
      lds_read_tr<g * 16 * 64 + j * KT * 64>(lo, vaddr[0]);
      lds_read_tr<g * 16 * 64 + j * KT * 64>(hi, vaddr[1]);
    };
    vs.prologue(rdv);                // lands under the whole QK^T + softmax phase
    // ---- S^T = ka . qa^T - m for the 32-key blocks of the tile ---------------------------
    f32x16_t s[2];
    RowStream<KSU * NKB, 4> ks_;
    auto rdk = [&](auto J, bf16x8_t& d) {
      constexpr int j = decltype(J)::value, kb = j / KSU, ks = j % KSU;
      lds_read128<kb * 2048 + (ks >> 1) * KT * 64>(d, kaddr[ks & 1]);
    };
    ks_.prologue(rdk);
    ks_.run(rdk, [&](auto J, const bf16x8_t& f) {
      constexpr int j = decltype(J)::value, kb = j / KSU, ks = j % KSU;
      s[kb] = mfma32(f, qf[ks], ks == 0 ? negm : s[kb]);
    });
    STAMP(8 + t * 8 + 2);
    const int kbase = t * KT;
    if (kbase + KT > a.Nk) {  // ragged last tile (uniform branch): rows >= Nk hold re-read data
      asm volatile("; ragged key rows" ::: "memory");
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbase + kb * 32 + acc_row(r, lane) >= a.Nk) s[kb][r] = -INFINITY;
    }
    // ---- online softmax: lane = one query, its two halves hold disjoint key rows; s is already
    // relative to the running maximum ------------------------------------------------------
    float mx;
    if constexpr (HALF) {
      mx = max3(s[0][0], s[0][1], s[0][2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = max3(mx, s[0][r], s[0][r + 1]);
      mx = fmaxf(mx, s[0][15]);
    } else {
      mx = max3(s[0][0], s[1][0], s[0][1]);
      mx = max3(mx, s[1][1], s[0][2]);
#pragma unroll
      for (int r = 2; r < 15; ++r) mx = max3(mx, s[1][r], s[0][r + 1]);
      mx = fmaxf(mx, s[1][15]);
    }
    mx = fmaxf(mx, other_half(mx));
    // defer-max: only re-base when the max grew by more than 2^RESCALE_THR; until then P is
    // bounded by 2^THR instead of 1, which fp32 accumulation absorbs (cdna guide T13).  The
    // previous tile's P.V is complete at this point, so O and l carry exactly one scale.  The
    // first tile always re-bases (m_run starts at 0, which may be far ABOVE every score).
    constexpr float RESCALE_THR = 6.0f;
    if (t == 0 || !__all(mx <= RESCALE_THR)) {
      asm volatile("; re-base" ::: "memory");
      const float shift = t == 0 ? mx : fmaxf(mx, 0.f);      // m_new - m_run
      if (t > 0) {
        const float alpha = fast_exp2(-shift);
        lacc[0] *= alpha;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[j][r] *= alpha;
      }
      m_run += shift;
#pragma unroll
      for (int r = 0; r < 16; ++r) negm[r] = -m_run;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] -= shift;
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = fast_exp2(s[kb][r]);
    STAMP(8 + t * 8 + 3);
    // ---- O^T += V^T . P^T (and l += 1^T P^T) ------------------------------------------------
    const bf16x8_t pf[4] = {acc_to_frag(s[0], 0), acc_to_frag(s[0], 1),
                            HALF ? acc_to_frag(s[0], 0) : acc_to_frag(s[1], 0),
                            HALF ? acc_to_frag(s[0], 1) : acc_to_frag(s[1], 1)};
    vs.run(rdv, [&](auto J, const bf16x8_t& f) {
      constexpr int i = decltype(J)::value, g = i / 3, j = i % 3;
      o[j] = mfma32(f, pf[g], o[j]);
      if constexpr (j == 2) lacc = mfma32(onesf, pf[g], lacc);
    });
    STAMP(8 + t * 8 + 4);
  };
  for (int t = 0; t + 1 < nt; ++t) tile(t, Int<0>{});
  if (a.Nk - (nt - 1) * KT > 32) tile(nt - 1, Int<0>{});
  else tile(nt - 1, Int<1>{});
#ifdef SVIT_ATTN_STAMPS
  if (stamp_on) { g_attn_stamps[2] = __builtin_readcyclecounter(); g_attn_stamps[3] = wall_clock64(); }
#endif

  // ---- epilogue: normalise, stage the wave's 32 x 96 tile in LDS, store whole rows with the
  // pooled query added (residual pooling), merge heads ---------------------------------------
  // residual-pooling operand (the pooled q rows, in the row-major chunk order of the output
  // stores): no LDS-DMA is in flight any more, plain loads are safe; they travel under the staging
  uint4 qres[6];
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int id = it * 64 + lane, row = id / 12, ch = id % 12;
    const int q = min(q0 + row, a.Nq - 1);
    if constexpr ((SVIT_ATTN_ABL & 2) != 0) qres[it] = make_uint4(0u, 0u, 0u, 0u);
    else qres[it] = *(const uint4*)(qa + (size_t)q * DA + ch * 8);
  }
  const float l_lo = __shfl(lacc[0], lane & 31, 64);   // row 0 of the sum block lives in lanes 0..31
  const float inv = 1.f / l_lo;
  if (hh == 0 && qi < a.Nq) a.lse2[(size_t)bh * a.Nq + qi] = m_run + log2f(l_lo);
  __builtin_amdgcn_s_barrier();        // every wave is done with the K/V ring
  constexpr int OROW = 208;            // bytes per staged row (192 + pad: spreads the banks)
  unsigned char* ost = smem + wave * (32 * OROW);
  {
    unsigned char* orow = ost + (lane & 31) * OROW;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dv = j * 32 + 8 * g + 4 * hh;
        uint2 pk;
        pk.x = pack_bf16x2(o[j][4 * g] * inv, o[j][4 * g + 1] * inv);
        pk.y = pack_bf16x2(o[j][4 * g + 2] * inv, o[j][4 * g + 3] * inv);
        *(uint2*)(orow + dv * 2) = pk;
      }
  }
  // 12 sixteen-byte chunks per row; lane -> (row, chunk): a row's 192 bytes are written by 12
  // consecutive lanes (the wave's own staging writes are ordered by the compiler's LDS waits)
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int id = it * 64 + lane, row = id / 12, ch = id % 12;
    const int q = q0 + row;
    if (q < a.Nq) {
      uint4 ov = *(const uint4*)(ost + row * OROW + ch * 16);
      if (q > 0) {                     // every token but cls adds its pooled q
        const uint4 qq = qres[it];
        ov.x = pack_bf16x2(lo_bf16(ov.x) + lo_bf16(qq.x), hi_bf16(ov.x) + hi_bf16(qq.x));
        ov.y = pack_bf16x2(lo_bf16(ov.y) + lo_bf16(qq.y), hi_bf16(ov.y) + hi_bf16(qq.y));
        ov.z = pack_bf16x2(lo_bf16(ov.z) + lo_bf16(qq.z), hi_bf16(ov.z) + hi_bf16(qq.z));
        ov.w = pack_bf16x2(lo_bf16(ov.w) + lo_bf16(qq.w), hi_bf16(ov.w) + hi_bf16(qq.w));
      }
      if constexpr ((SVIT_ATTN_ABL & 4) != 0) { if (ov.x == 0x12345678u && a.Nk < 0) a.lse2[0] = 1.f; }
      else *(uint4*)((bf16_t*)a.ctx + ((size_t)b * a.Nq + q) * a.heads * HD + head * HD + ch * 8) = ov;
    }
  }
#ifdef SVIT_ATTN_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  if (stamp_on) g_attn_stamps[5] = __builtin_readcyclecounter();
#endif
#endif
}


// ---------------------------------------------------------------------------------------
// T' = 1 tile (round 4): Nk <= 64 -- the single-frame passes (the reference's frames pass, tools/train_net.py:105-110, and
// the image ranks: 7x7 pooled keys + cls + objects = 54 at every block).  One 64-key tile is the WHOLE key range, so there
// is no ring, no online re-base and nothing for the waves of a workgroup to meet about after the K / V image has landed:
//   * K and V of the (batch, head) arrive ONCE per workgroup (LDS-DMA, clamped rows past Nk) behind one barrier;
//   * a workgroup is persistent over `tiles_per_wg` 128-query tiles; every wave walks its own 32-row sub-tiles with NO
//     further barrier: Q fragments by row-per-lane loads issued ONE TILE AHEAD (the latency the generic kernel's prologue
//     exposes per workgroup), scores, one exp2 per score relative to the tile's maximum, P.V, the row sum on the matrix
//     pipe, output through a wave-private LDS region as whole 192-byte rows (+ the pooled-q residual);
//   * the arithmetic is the generic kernel's first tile, operation for operation: results are bit-identical to it
//     (tests/test_kernels_gpu.py::test_attention_fwd_short_key_tile).
template <int KSU>
__global__ __launch_bounds__(256, 2) void attn_fwd_short_kernel(svit_attn_fwd_args a, int tiles_per_wg) {
#if __HIP_DEVICE_COMPILE__
  constexpr int NP = (KSU + 1) / 2, KCOLS = NP * 32;
  constexpr int K_BYTES = KT * KCOLS * 2, V_BYTES = KT * HD * 2, STAGE = K_BYTES + V_BYTES;
  using KLoad = BufTile<KT, KCOLS, 4>;
  using VLoad = BufTile<KT, HD, 4>;
  constexpr int OROW = 208;            // bytes per staged output row (192 + pad: spreads the banks)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int DA = a.DA;
  const int wgid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int bh = wgid / gridDim.x, b = bh / a.heads, head = bh % a.heads;
  const int ntiles = (a.Nq + 127) / 128;
  const int t_begin = (wgid % gridDim.x) * tiles_per_wg, t_end = min(ntiles, t_begin + tiles_per_wg);
  const bf16_t* qa = (const bf16_t*)a.qa + ((size_t)bh * a.Nq) * DA;
  const bf16_t* ka = (const bf16_t*)a.ka + ((size_t)bh * a.Nk) * DA;
  const bf16_t* vv = (const bf16_t*)a.v + ((size_t)bh * a.Nk) * HD;

  KLoad kload;
  VLoad vload;
  kload.init(DA, wave, lane);
  vload.init(HD, wave, lane);
  const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)ka, 0, a.Nk * DA * 2, 0x00020000);
  const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vv, 0, a.Nk * HD * 2, 0x00020000);
  kload.issue_auto(krs, 0u, DA, a.Nk, smem, wave, lane);
  vload.issue_auto(vrs, 0u, HD, a.Nk, smem + K_BYTES, wave, lane);
  // the wave's Q fragments of its first tile (lane -> its own row, 16 bytes per k-step; rows past Nq re-read the last)
  bf16x8_t qn[KSU];
  auto load_q = [&](int tile) {
    const int qc = min(tile * 128 + wave * 32 + (lane & 31), a.Nq - 1);
#pragma unroll
    for (int ks = 0; ks < KSU; ++ks) qn[ks] = *(const bf16x8_t*)(qa + (size_t)qc * DA + ks * 16 + hh * 8);
  };
  load_q(t_begin);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();        // the K / V image is complete; the only meeting point of the workgroup

  bf16x8_t onesf;                      // A operand of the row-sum MFMA (row 0 = sum over the 16 keys of a P fragment)
#pragma unroll
  for (int e = 0; e < 8; ++e) onesf[e] = (lane & 31) == 0 ? (__bf16)1.0f : (__bf16)0.0f;
  const unsigned lds0 = (unsigned)(size_t)smem;
  unsigned kaddr[2], vaddr[2];
  {
    const int row = lane & 31, sw = (row >> 2) & 3;
    kaddr[0] = lds0 + row * 64 + 16 * ((0 + hh) ^ sw);
    kaddr[1] = lds0 + row * 64 + 16 * ((2 + hh) ^ sw);
    const int cg = (lane >> 4) & 1, i = lane & 15, q = i >> 2, pp = i & 3;
    const int r0 = 4 * hh, ch = 2 * cg + (pp >> 1);
    const unsigned vb = lds0 + K_BYTES + 8 * (pp & 1);
    vaddr[0] = vb + (r0 + q) * 64 + 16 * (ch ^ ((r0 >> 2) & 3));
    vaddr[1] = vb + (r0 + 8 + q) * 64 + 16 * (ch ^ (((r0 + 8) >> 2) & 3));
  }
  unsigned char* ost = smem + STAGE + wave * (32 * OROW);      // wave-private output staging

  auto walk = [&](auto HalfTag) {
    constexpr bool HALF = decltype(HalfTag)::value == 1;       // Nk <= 32: only key block 0 is multiplied
    constexpr int NKB = HALF ? 1 : 2;
    for (int tile = t_begin; tile < t_end; ++tile) {
      const int q0 = tile * 128 + wave * 32;
      if (q0 >= a.Nq) break;                                   // (wave-uniform: this wave has no rows left)
      const int qi = q0 + (lane & 31);
      bf16x8_t qf[KSU];
#pragma unroll
      for (int ks = 0; ks < KSU; ++ks) qf[ks] = qn[ks];
#pragma unroll
      for (int ks = 0; ks < KSU; ++ks) asm volatile("" : "+v"(qf[ks]));
      if (tile + 1 < t_end) load_q(tile + 1);                  // travels under this tile
      // residual-pooling operand (the pooled q rows, in the row-major chunk order of the output stores)
      uint4 qres[6];
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int id = it * 64 + lane, row = id / 12, ch = id % 12;
        const int q = min(q0 + row, a.Nq - 1);
        qres[it] = *(const uint4*)(qa + (size_t)q * DA + ch * 8);
      }
      TrStream<6 * NKB, 3> vs;
      auto rdv = [&](auto J, s16x4_t& lo, s16x4_t& hi) {
        constexpr int i = decltype(J)::value, g = i / 3, j = i % 3;
        lds_read_tr<g * 16 * 64 + j * KT * 64>(lo, vaddr[0]);
        lds_read_tr<g * 16 * 64 + j * KT * 64>(hi, vaddr[1]);
      };
      vs.prologue(rdv);
      f32x16_t s[2], zero;
#pragma unroll
      for (int r = 0; r < 16; ++r) zero[r] = 0.f;
      RowStream<KSU * NKB, 4> ks_;
      auto rdk = [&](auto J, bf16x8_t& d) {
        constexpr int j = decltype(J)::value, kb = j / KSU, ks = j % KSU;
        lds_read128<kb * 2048 + (ks >> 1) * KT * 64>(d, kaddr[ks & 1]);
      };
      ks_.prologue(rdk);
      ks_.run(rdk, [&](auto J, const bf16x8_t& f) {
        constexpr int j = decltype(J)::value, kb = j / KSU, ks = j % KSU;
        s[kb] = mfma32(f, qf[ks], ks == 0 ? zero : s[kb]);
      });
      if (KT > a.Nk) {       // rows >= Nk hold re-read data (uniform branch)
        asm volatile("; ragged key rows" ::: "memory");
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kb * 32 + acc_row(r, lane) >= a.Nk) s[kb][r] = -INFINITY;
      }
      float mx;
      if constexpr (HALF) {
        mx = max3(s[0][0], s[0][1], s[0][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = max3(mx, s[0][r], s[0][r + 1]);
        mx = fmaxf(mx, s[0][15]);
      } else {
        mx = max3(s[0][0], s[1][0], s[0][1]);
        mx = max3(mx, s[1][1], s[0][2]);
#pragma unroll
        for (int r = 2; r < 15; ++r) mx = max3(mx, s[1][r], s[0][r + 1]);
        mx = fmaxf(mx, s[1][15]);
      }
      mx = fmaxf(mx, other_half(mx));
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = fast_exp2(s[kb][r] - mx);
      const bf16x8_t pf[4] = {acc_to_frag(s[0], 0), acc_to_frag(s[0], 1),
                              HALF ? acc_to_frag(s[0], 0) : acc_to_frag(s[1], 0),
                              HALF ? acc_to_frag(s[0], 1) : acc_to_frag(s[1], 1)};
      f32x16_t o[3], lacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) lacc[r] = 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
      vs.run(rdv, [&](auto J, const bf16x8_t& f) {
        constexpr int i = decltype(J)::value, g = i / 3, j = i % 3;
        o[j] = mfma32(f, pf[g], o[j]);
        if constexpr (j == 2) lacc = mfma32(onesf, pf[g], lacc);
      });
      // ---- normalise, stage the 32 x 96 tile, store whole rows with the pooled query added -------------------------
      const float l_lo = __shfl(lacc[0], lane & 31, 64);
      const float inv = 1.f / l_lo;
      if (hh == 0 && qi < a.Nq) a.lse2[(size_t)bh * a.Nq + qi] = mx + log2f(l_lo);
      {
        unsigned char* orow = ost + (lane & 31) * OROW;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int dv = j * 32 + 8 * g + 4 * hh;
            uint2 pk;
            pk.x = pack_bf16x2(o[j][4 * g] * inv, o[j][4 * g + 1] * inv);
            pk.y = pack_bf16x2(o[j][4 * g + 2] * inv, o[j][4 * g + 3] * inv);
            *(uint2*)(orow + dv * 2) = pk;
          }
      }
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int id = it * 64 + lane, row = id / 12, ch = id % 12;
        const int q = q0 + row;
        if (q < a.Nq) {
          uint4 ov = *(const uint4*)(ost + row * OROW + ch * 16);
          if (q > 0) {
            const uint4 qq = qres[it];
            ov.x = pack_bf16x2(lo_bf16(ov.x) + lo_bf16(qq.x), hi_bf16(ov.x) + hi_bf16(qq.x));
            ov.y = pack_bf16x2(lo_bf16(ov.y) + lo_bf16(qq.y), hi_bf16(ov.y) + hi_bf16(qq.y));
            ov.z = pack_bf16x2(lo_bf16(ov.z) + lo_bf16(qq.z), hi_bf16(ov.z) + hi_bf16(qq.z));
            ov.w = pack_bf16x2(lo_bf16(ov.w) + lo_bf16(qq.w), hi_bf16(ov.w) + hi_bf16(qq.w));
          }
          *(uint4*)((bf16_t*)a.ctx + ((size_t)b * a.Nq + q) * a.heads * HD + head * HD + ch * 8) = ov;
        }
      }
    }
  };
  if (a.Nk > 32) walk(Int<0>{});
  else walk(Int<1>{});
#endif
}

#ifndef SVIT_ATTN_SHORT_WGS     // workgroups the T' = 1 launch aims at (swept inside the frames-pass step in round 4)
#define SVIT_ATTN_SHORT_WGS 512
#endif
template <int KSU>
int launch_short(const svit_attn_fwd_args& a, hipStream_t st) {
  constexpr int NP = (KSU + 1) / 2;
  const size_t lds = (size_t)(KT * NP * 32 * 2 + KT * HD * 2) + (size_t)4 * 32 * 208;
  static SvitOnce once;
  if (int rc = svit_max_lds_once(once, (const void*)attn_fwd_short_kernel<KSU>, lds)) return rc;
  // ~two workgroups per CU in one round; a workgroup walks its share of the (batch, head)'s 128-query tiles
  constexpr int want = SVIT_ATTN_SHORT_WGS;
  const int ntiles = (a.Nq + 127) / 128, bh = a.B * a.heads;
  int chunks = (want + bh - 1) / bh;
  chunks = std::max(1, std::min(chunks, ntiles));
  const int per = (ntiles + chunks - 1) / chunks;
  chunks = (ntiles + per - 1) / per;
  hipLaunchKernelGGL((attn_fwd_short_kernel<KSU>), dim3(chunks, bh), dim3(256), lds, st, a, per);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

template <int KSU, int NW, int NS>
int launch_cfg(const svit_attn_fwd_args& a, hipStream_t st) {
  constexpr int NP = (KSU + 1) / 2;
  const size_t stage = (size_t)(KT * NP * 32 * 2 + KT * HD * 2);
  size_t lds = NS * stage;
  const size_t lds_out = (size_t)NW * 32 * 208;
  if (lds < lds_out) lds = lds_out;
  if (SVIT_ATTN_QSTAGE)      // the Q staging region behind the prologue's stages (RowStage, 8 / 11 KiB per wave)
    lds = std::max(lds, (NS - 1) * stage + NW * (size_t)attn::RowStage<(KSU <= 8 ? 128 : 160)>::BYTES);
  static SvitOnce once;
  if (int rc = svit_max_lds_once(once, (const void*)attn_fwd_kernel<KSU, NW, NS>, lds)) return rc;
  dim3 grid((a.Nq + NW * 32 - 1) / (NW * 32), a.B * a.heads);
  hipLaunchKernelGGL((attn_fwd_kernel<KSU, NW, NS>), grid, dim3(NW * 64), lds, st, a);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

template <int KSU>
int launch_fwd(const svit_attn_fwd_args& a, hipStream_t st) {
  // 8-wave workgroups once they still cover the chip (one per CU)
  // measured (tools/bench_kernels.py attn): the 8-wave form wins 3-4 % on the long-key blocks
  // (Nk = 1633, DA = 160) and loses on the short ones, where the 8-wave barrier dominates
  const long wg8 = (long)((a.Nq + 255) / 256) * a.B * a.heads;
  const bool wide = a.DA == 160 && wg8 >= 200;
  return wide ? launch_cfg<KSU, 8, 3>(a, st) : launch_cfg<KSU, 4, 2>(a, st);
}
}  // namespace
// (round 6: an 8-wave ANTI-PHASE forward -- halves of the workgroup alternating a matrix segment [P.V(t-1) ; QK^T(t)] and a vector
// segment [softmax(t) ; LDS-DMA issue] between barriers, K / V rings four tiles deep -- was built, parity-tested and measured 0.95-0.98
// of these kernels on its best shapes, 0.67-0.92 elsewhere: its vector segment (five LDS-DMA pieces at ~140 cycles of issue each beside a
// partner streaming MFMAs) is longer than its matrix segment, and the form pays two barriers per tile.  Anatomy by ablation:
// profiles/r06_attn_anti_phase.txt; the kernel, its tests and tools: tools/diag/variants/attn_fwd_anti_phase.patch)
// (the one-wave-per-SIMD, 64-rows-per-wave forward of round 4 measured 1.5x slower on every shape of the model --
// profiles/r04_attn_w64.txt -- and lives on as tools/diag/variants/attn_fwd64.hip, no longer part of the library)
extern "C" int svit_attn_fwd(const svit_attn_fwd_args* a, void* stream) {
  if (!a || !a->qa || !a->ka || !a->v || !a->ctx || !a->lse2) return SVIT_ERR_ARG;
  if (a->B <= 0 || a->heads <= 0 || a->Nq <= 0 || a->Nk <= 0) return SVIT_ERR_SHAPE;
  if (((uintptr_t)a->qa | (uintptr_t)a->ka | (uintptr_t)a->v | (uintptr_t)a->ctx) & 15)
    return SVIT_ERR_ALIGN;
  if (a->DA != 128 && a->DA != 160) return SVIT_ERR_SHAPE;
  if (a->bias_cols < 0 || a->bias_cols > a->DA - 96) return SVIT_ERR_ARG;
  // (a one-wave-per-SIMD software-pipelined form was built and measured 27-36 % slower -- a single
  // wave per SIMD reaches half the VALU issue rate; it lives on as tools/diag/attn_fwd2_experiment.hip)
  const int extra = a->DA - 96;
  const int bias_cols = a->bias_cols > 0 ? a->bias_cols : extra;
  // round 4: the T' = 1 tile -- the whole key range is one 64-key tile (frames pass, image ranks)
  if (a->Nk <= KT && svit_knob(SVIT_K_ATTN_FWD_SHORT) != 0) {
    if (6 + (bias_cols + 15) / 16 == 7) return launch_short<7>(*a, (hipStream_t)stream);
    if (6 + (bias_cols + 15) / 16 == 8) return launch_short<8>(*a, (hipStream_t)stream);
  }
  switch (6 + (bias_cols + 15) / 16) {
    case 7: return launch_fwd<7>(*a, (hipStream_t)stream);
    case 8: return launch_fwd<8>(*a, (hipStream_t)stream);
    case 9: return launch_fwd<9>(*a, (hipStream_t)stream);
    case 10: return launch_fwd<10>(*a, (hipStream_t)stream);
  }
  return SVIT_ERR_SHAPE;
}
