// Fused pooled attention, backward (flash-style recompute) -- gfx950.
//
// Two launches (both on the caller's stream):
//   1. dq kernel : first delta[q] = sum_c dctx[q,c] * (ctx[q,c] - q_pooled[q,c]) (rowsum(dO*O)),
//                  kept in a register and stored as (-lse2, -delta) planes in the caller's scratch
//                  for launch 2; then per 32-query wave, sweep K/V tiles: S^T, P^T = exp2(S - lse2)
//                  (S = qa . ka^T is the score in the log2 domain: the keys carry scale * log2 e),
//                  dP^T = V dO^T, dS^T = P^T (dP^T - delta), dQa^T += Ka^T dS^T (x ln 2 at the end).
//                  Query on the lane => P/dS reach the next MFMA as B operands in registers.
//   2. dkv kernel: per 32-key wave (128 keys per block), sweep 64-query stages of a query
//                  chunk: S, P, dP, dS with the KEY on the lane; dV^T += dO^T P, dK^T += Q^T dS
//                  accumulate in registers over the whole sweep; chunks of the query range run
//                  in different blocks, each of which stores its partial dK / dV into its own
//                  plane dk[part] / dv[part] with plain stores straight from the accumulator
//                  registers; the consumer (svit_pool_ln_bwd, main_parts) adds the planes while it
//                  reads them.  (Rounds 1-2 met in fp32 atomics behind an LDS transpose: in-kernel
//                  stamps put that epilogue at 7-14 us of a 40 us launch.)  Bit-reproducible.
// No N x N matrix is ever stored.  The residual-pooling path (ctx += q) contributes dctx to dq
// outside these kernels (svit_pool_ln_bwd's d_res input).
//
// Round 3: every LDS fragment read is issued by hand (attn_common.h RowStream / TrStream), four
// fragments ahead of the MFMA that consumes it.  The round-2 kernels left the reads to hipcc, which
// (a) waited lgkmcnt(0) in front of almost every MFMA and (b) in the dkv kernel put an
// `s_waitcnt vmcnt(0)` in front of the plain LDS loads of the (lse2, delta) pairs, i.e. drained the
// LDS-DMA ring in the middle of every tile.  Also new: only the k-steps of the QK^T contraction that
// carry data are multiplied (bias_cols, as in the forward); the per-row constants enter as the
// initial accumulators of the S and dP chains (-lse2 and -delta: in the dkv kernel read from LDS
// straight into the accumulator registers, in the dq kernel two constant register blocks at DA = 128)
// so that P and dS cost two VALU operations per element instead of five (one v_exp, one v_mul: the
// score arrives in the log2 domain, attn_fwd.hip); ln 2 / `scale` are applied once to the finished
// dQ / dK tiles; the LDS-DMA pieces are
// addressed by buffer descriptor + scalar offset (no per-piece address arithmetic); the dkv kernel
// consumes 64 queries per barrier (was 32); dQa leaves through LDS as whole rows in 16-byte stores.
#include <algorithm>
#include "attn_common.h"
#include "../../include/svit_hip.h"

#ifdef SVIT_ATTN_STAMPS   // tools/attn_bwd_stamps.py: wall-clock (100 MHz) stamps of workgroup 0 / wave 0
__device__ unsigned long long g_bwd_stamps[64];
extern "C" int svit_debug_attn_bwd_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bwd_stamps), sizeof(unsigned long long) * n);
}
#define BSTAMP(slot)                                                                    \
  do {                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    if (stamp_on) g_bwd_stamps[(slot)] = wall_clock64();                                \
    __builtin_amdgcn_sched_barrier(0);                                                  \
  } while (0)
#else
#define BSTAMP(slot) do {} while (0)
#endif

#ifndef SVIT_ATTN_QSTAGE      // 0 in a diagnostic build: the round-3 row-per-lane Q loads (A/B)
#define SVIT_ATTN_QSTAGE 1
#endif

namespace {
using namespace attn;
constexpr int KT = 64;   // keys per tile (dq kernel)
constexpr int QR = 64;   // queries per stage (dkv kernel)
constexpr float LN2 = 0.6931471805599453f;

// per-lane LDS byte offsets of the fragment reads inside a 64-row panel image (attn_common.h):
// row fragments of k-step ks of the 32-row block at row0 sit at rowa[ks&1] + row0*64 + (ks>>1)*4096,
// the transposed fragment (lo, hi) of the 16 rows at rbase / panel p at tra[0|1] + rbase*64 + p*4096.
struct FragAddr {
  unsigned rowa[2], tra[2];
  __device__ __forceinline__ void init(unsigned base, int lane) {
    const int hh = lane >> 5, row = lane & 31, sw = (row >> 2) & 3;
    rowa[0] = base + row * 64 + 16 * ((0 + hh) ^ sw);
    rowa[1] = base + row * 64 + 16 * ((2 + hh) ^ sw);
    const int cg = (lane >> 4) & 1, i = lane & 15, q = i >> 2, pp = i & 3;
    const int r0 = 4 * hh, ch = 2 * cg + (pp >> 1);
    const unsigned tb = base + 8 * (pp & 1);
    tra[0] = tb + (r0 + q) * 64 + 16 * (ch ^ ((r0 >> 2) & 3));
    tra[1] = tb + (r0 + 8 + q) * 64 + 16 * (ch ^ (((r0 + 8) >> 2) & 3));
  }
};

// ---------------------------------------------------------------------------------------
// dq kernel.  K/V tiles travel HBM -> LDS by LDS-DMA (no staging registers), two stages, one
// raw barrier per tile behind the wave's vmcnt wait; 2 blocks per CU (2 waves per SIMD) so one
// wave's exp/convert VALU work overlaps the other's MFMAs.  KSU = k-steps of the S contraction
// that carry data (6 + ceil(bias columns / 16)).
template <int DA, int KSU>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(svit_attn_bwd_args a) {
#if __HIP_DEVICE_COMPILE__
  constexpr int NP = DA / 32;
  constexpr int K_BYTES = KT * DA * 2, V_BYTES = KT * HD * 2, STAGE = K_BYTES + V_BYTES;
  using KLoad = BufTile<KT, DA, 4>;
  using VLoad = BufTile<KT, HD, 4>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
#ifdef SVIT_ATTN_STAMPS
  const bool stamp_on = blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
#endif
  BSTAMP(0);
  const int wgid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int bh = wgid / gridDim.x, b = bh / a.heads, head = bh % a.heads;
  const int qtile = wgid % gridDim.x;
  const int q0 = qtile * 128 + wave * 32;
  const int qi = q0 + (lane & 31);
  const int qc = min(qi, a.Nq - 1);
  const bf16_t* qa = (const bf16_t*)a.qa + ((size_t)bh * a.Nq) * DA;
  const bf16_t* ka = (const bf16_t*)a.ka + ((size_t)bh * a.Nk) * DA;
  const bf16_t* vv = (const bf16_t*)a.v + ((size_t)bh * a.Nk) * HD;
  const bf16_t* dor = (const bf16_t*)a.dctx + ((size_t)b * a.Nq + qc) * a.heads * HD + head * HD;
  const float lse = a.lse2[(size_t)bh * a.Nq + qc];

  const int nt = (a.Nk + KT - 1) / KT;
  KLoad kload;
  VLoad vload;
  kload.init(DA, wave, lane);
  vload.init(HD, wave, lane);
  const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)ka, 0, a.Nk * DA * 2, 0x00020000);
  const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vv, 0, a.Nk * HD * 2, 0x00020000);
  auto issue = [&](int t) {
    unsigned char* st = smem + (t & 1) * STAGE;
    const unsigned k0 = (unsigned)t * KT;
    kload.issue_auto(krs, k0 * DA * 2u, DA, a.Nk - (int)k0, st, wave, lane);
    vload.issue_auto(vrs, k0 * HD * 2u, HD, a.Nk - (int)k0, st + K_BYTES, wave, lane);
  };
  BSTAMP(1);
  issue(0);                          // travels while the register operands are fetched
  BSTAMP(2);

  bf16x8_t qf[KSU], dof[6];
  // (round 4) the wave's 32 Q rows come through LDS, coalesced (attn_common.h RowStage; by ablation the row-per-lane
  // loads of the forward's Q cost 1.9 us of a 20.7 us launch), into stage 1 and the tail of the allocation, free
  // until tile 1 is issued behind the first barrier; dO and O stay row-per-lane loads (no room for three regions
  // beside stage 0 at two workgroups per CU) and are requested first
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) dof[ks] = *(const bf16x8_t*)(dor + ks * 16 + hh * 8);
  using QStage = RowStage<DA>;
  if constexpr (SVIT_ATTN_QSTAGE != 0) {
    unsigned char* qreg = smem + STAGE + wave * QStage::BYTES;
    QStage::issue(qa, DA, q0, a.Nq, qreg, lane);
    wait_vmcnt<0>();
#pragma unroll
    for (int ks = 0; ks < KSU; ++ks) qf[ks] = QStage::frag(qreg, ks, lane);
  } else {
#pragma unroll
    for (int ks = 0; ks < KSU; ++ks) qf[ks] = *(const bf16x8_t*)(qa + (size_t)qc * DA + ks * 16 + hh * 8);
  }
  // delta = rowsum(dO . O) with O = ctx - q (residual pooling adds q to every token but cls):
  // each half-wave lane holds 48 of the 96 channels.  (-lse2, -delta) are what the dkv kernel
  // loads as the initial accumulators of its S and dP chains, so they are written in that form.
  float dlt;
  {
    const bf16_t* orow = (const bf16_t*)a.ctx + ((size_t)b * a.Nq + qc) * a.heads * HD + head * HD;
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
      const uint4 o = *(const uint4*)(orow + ks * 16 + hh * 8);
      const uint4 d = __builtin_bit_cast(uint4, dof[ks]);
      uint4 q = __builtin_bit_cast(uint4, qf[ks]);
      if (qc == 0) q = make_uint4(0, 0, 0, 0);
      part += lo_bf16(d.x) * (lo_bf16(o.x) - lo_bf16(q.x)) + hi_bf16(d.x) * (hi_bf16(o.x) - hi_bf16(q.x));
      part += lo_bf16(d.y) * (lo_bf16(o.y) - lo_bf16(q.y)) + hi_bf16(d.y) * (hi_bf16(o.y) - hi_bf16(q.y));
      part += lo_bf16(d.z) * (lo_bf16(o.z) - lo_bf16(q.z)) + hi_bf16(d.z) * (hi_bf16(o.z) - hi_bf16(q.z));
      part += lo_bf16(d.w) * (lo_bf16(o.w) - lo_bf16(q.w)) + hi_bf16(d.w) * (hi_bf16(o.w) - hi_bf16(q.w));
    }
    dlt = part + __shfl_xor(part, 32, 64);
    if (hh == 0 && qi < a.Nq) {
      float* sc = a.delta + (size_t)bh * a.Nq * 2;
      sc[qi] = -lse;
      sc[a.Nq + qi] = -dlt;
    }
  }

  // Pin the register operands NOW: their first use must not sit inside the tile loop, or the
  // compiler's wait for them (vmcnt(0)) would drain the LDS-DMA pipeline every iteration.
#pragma unroll
  for (int ks = 0; ks < KSU; ++ks) asm volatile("" : "+v"(qf[ks]));
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) asm volatile("" : "+v"(dof[ks]));
  float lse_p = lse, dlt_p = dlt;
  asm volatile("" : "+v"(lse_p), "+v"(dlt_p));

  f32x16_t dq[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[j][r] = 0.f;

  FragAddr fa;
  fa.init((unsigned)(size_t)smem, lane);
  // initial accumulators of the S and dP chains: -lse2 and -delta of this lane's query in all 16
  // registers where the register budget allows it (DA = 128), zeros otherwise
  constexpr bool CINIT = DA == 128;
  f32x16_t s_init, dp_init;
#pragma unroll
  for (int r = 0; r < 16; ++r) { s_init[r] = CINIT ? -lse_p : 0.f; dp_init[r] = CINIT ? -dlt_p : 0.f; }

  static_prio(blockIdx.y * gridDim.x + blockIdx.x, wave, 4);
  BSTAMP(3);
  for (int t = 0; t < nt; ++t) {
    if (t < 4) BSTAMP(8 + 4 * t);
    wait_vmcnt<0>();                 // this wave's share of tile t has landed
    if (t < 4) BSTAMP(9 + 4 * t);
    __builtin_amdgcn_s_barrier();    // everyone's share has; everyone is done with tile t-1
    if (t < 4) BSTAMP(10 + 4 * t);
    if (t + 1 < nt) issue(t + 1);    // travels while tile t is consumed
    if (t < 4) BSTAMP(11 + 4 * t);
    const unsigned so = (t & 1) * STAGE;
    const unsigned ra[2] = {fa.rowa[0] + so, fa.rowa[1] + so};
    const unsigned ta[2] = {fa.tra[0] + so, fa.tra[1] + so};
    const int kbase = t * KT;
    static_for<0, 2>([&](auto KB) {
      constexpr int kb = decltype(KB)::value;
      f32x16_t s, dp;
      // ---- S^T (KSU k-steps of K rows) and dP^T (6 k-steps of V rows) ------------------------
      RowStream<KSU + 6, 4> rs;
      auto rd = [&](auto J, bf16x8_t& d) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < KSU) lds_read128<kb * 2048 + (j >> 1) * 4096>(d, ra[j & 1]);
        else lds_read128<K_BYTES + kb * 2048 + ((j - KSU) >> 1) * 4096>(d, ra[(j - KSU) & 1]);
      };
      rs.prologue(rd);
      rs.run(rd, [&](auto J, const bf16x8_t& f) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < KSU) s = mfma32(f, qf[j], j == 0 ? s_init : s);
        else dp = mfma32(f, dof[j - KSU], j == KSU ? dp_init : dp);
      });
      // ---- K^T fragments of the dQ product: the first ones travel under the VALU section -----
      TrStream<2 * NP, 3> ts;
      auto rdt = [&](auto J, s16x4_t& lo, s16x4_t& hi) {
        constexpr int j = decltype(J)::value, sp = j / NP, p = j % NP;
        lds_read_tr<(kb * 32 + sp * 16) * 64 + p * 4096>(lo, ta[0]);
        lds_read_tr<(kb * 32 + sp * 16) * 64 + p * 4096>(hi, ta[1]);
      };
      ts.prologue(rdt);
      if (kbase + KT > a.Nk) {   // ragged last tile (uniform branch): rows past Nk hold re-read data
        asm volatile("; ragged key rows" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbase + kb * 32 + acc_row(r, lane) >= a.Nk) s[r] = -INFINITY;
      }
      if constexpr (CINIT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r]) * dp[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r] - lse_p) * (dp[r] - dlt_p);
      }
      const bf16x8_t dsf0 = acc_to_frag(s, 0), dsf1 = acc_to_frag(s, 1);
      ts.run(rdt, [&](auto J, const bf16x8_t& f) {
        constexpr int j = decltype(J)::value, sp = j / NP, p = j % NP;
        dq[p] = mfma32(f, sp == 0 ? dsf0 : dsf1, dq[p]);
      });
    });
  }

  BSTAMP(4);
  // ---- epilogue: x ln 2, stage the wave's 32 x DA tile in LDS, store whole rows --------------
  __builtin_amdgcn_s_barrier();        // every wave is done with the K/V ring
  constexpr int OROW = DA * 2 + 16;    // bytes per staged row (pad spreads the banks)
  unsigned char* ost = smem + wave * (32 * OROW);
  // ---- fold mode (relD + relR, no relX, narrow tables): the rel-pos backward's query-side work happens HERE,
  // from the accumulators, before anything is staged.  The lane owns a query: its d(relq) values (the extra
  // column blocks of dq) are scattered into the wave's own 32 x ldd D rows in LDS, D . R^T is multiplied on the
  // matrix pipe (table fragments straight from L2) in the accumulator layout of dq, and ADDED to dq's first
  // 96 columns -- dqa then is the whole gradient of the pooled q and no dq_extra tensor exists.
  const bool fold = a.relD && a.relR && !a.relX && a.relD_ld <= 128;
  if (fold) {
    constexpr int EXTRA_ = DA - 96;
    unsigned char* dst = smem + 4 * (32 * OROW) + wave * (32 * 128 * 2);
    const int ldd = a.relD_ld, cpr = ldd >> 3, q = lane & 31;
    for (int i = lane; i < 32 * cpr; i += 64) *(uint4*)(dst + i * 16) = make_uint4(0u, 0u, 0u, 0u);
    int4 mp[NP - 3][4];
#pragma unroll
    for (int pj = 0; pj < NP - 3; ++pj)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        mp[pj][g] = q0 + q < a.Nq ? *(const int4*)(a.relD_map + (size_t)(q0 + q) * EXTRA_ + pj * 32 + 8 * g + 4 * hh)
                                  : make_int4(-1, -1, -1, -1);
    // the table fragments of the first k-step are requested together with the map: both round trips overlap,
    // and every later k-step's fragments fly while the MFMAs of the step before run
    const bf16_t* R = (const bf16_t*)a.relR;
    bf16x8_t rcur[3], rnxt[3];
#pragma unroll
    for (int pb = 0; pb < 3; ++pb) rcur[pb] = *(const bf16x8_t*)(R + (size_t)(pb * 32 + q) * ldd + 8 * hh);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int pj = 0; pj < NP - 3; ++pj)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cc[4] = {mp[pj][g].x, mp[pj][g].y, mp[pj][g].z, mp[pj][g].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // the two roundings of the unfused pair: dqa's bf16 column, then the scattered value
          const bf16_t v1 = f32_to_bf16(dq[3 + pj][4 * g + e] * LN2);
          if (cc[e] >= 0) *(bf16_t*)(dst + (q * ldd + cc[e]) * 2) = f32_to_bf16(bf16_to_f32(v1) * a.relD_scale);
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x16_t ex[3];
#pragma unroll
    for (int pb = 0; pb < 3; ++pb)
#pragma unroll
      for (int r = 0; r < 16; ++r) ex[pb][r] = 0.f;
    const int nks = ldd / 16;
    for (int ks = 0; ks < nks; ++ks) {
      if (ks + 1 < nks) {
#pragma unroll
        for (int pb = 0; pb < 3; ++pb)
          rnxt[pb] = *(const bf16x8_t*)(R + (size_t)(pb * 32 + q) * ldd + 16 * (ks + 1) + 8 * hh);
      }
      const bf16x8_t dfrag = *(const bf16x8_t*)(dst + (q * ldd + 16 * ks + 8 * hh) * 2);
#pragma unroll
      for (int pb = 0; pb < 3; ++pb) ex[pb] = mfma32(rcur[pb], dfrag, ex[pb]);
#pragma unroll
      for (int pb = 0; pb < 3; ++pb) rcur[pb] = rnxt[pb];
    }
    constexpr float INV_LN2 = 1.4426950408889634f;      // dq is in log2 units until the x ln 2 below
#pragma unroll
    for (int pb = 0; pb < 3; ++pb)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[pb][r] += ex[pb][r] * INV_LN2;
    bf16_t* Dg = (bf16_t*)a.relD + ((size_t)bh * a.Nq) * ldd;
    for (int i = lane; i < 32 * cpr; i += 64) {
      const int rl = i / cpr, ch = i % cpr;
      if (q0 + rl < a.Nq) *(uint4*)(Dg + (size_t)(q0 + rl) * ldd + ch * 8) = *(const uint4*)(dst + i * 16);
    }
  }
  {
    unsigned char* orow = ost + (lane & 31) * OROW;
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack_bf16x2(dq[j][4 * g] * LN2, dq[j][4 * g + 1] * LN2);
        pk.y = pack_bf16x2(dq[j][4 * g + 2] * LN2, dq[j][4 * g + 3] * LN2);
        *(uint2*)(orow + (j * 32 + 8 * g + 4 * hh) * 2) = pk;
      }
  }
  // (relD, below) the map entries of this wave's rows: requested now, so that their round trip overlaps the
  // row stores of dqa instead of standing in front of the scatter
  constexpr int EXTRA = DA - 96, NV = 32 * EXTRA / 64;
  int cols[NV];
  if (a.relD && !fold) {
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int e = it * 64 + lane, rl = e / EXTRA, j = e % EXTRA;
      cols[it] = q0 + rl < a.Nq ? a.relD_map[(size_t)(q0 + rl) * EXTRA + j] : -1;
    }
  }
  constexpr int CPR = DA / 8;          // 16-byte chunks per row
  bf16_t* out = (bf16_t*)a.dqa + ((size_t)bh * a.Nq) * DA;
#pragma unroll
  for (int it = 0; it < CPR / 2; ++it) {
    const int id = it * 64 + lane, row = id / CPR, ch = id % CPR;
    if (q0 + row < a.Nq)
      *(uint4*)(out + (size_t)(q0 + row) * DA + ch * 8) = *(const uint4*)(ost + row * OROW + ch * 16);
  }
  // ---- optional: the rel-pos backward's scattered matrix D, rows of this wave (svit_attn_bwd_args.relD).
  // The staged tile still holds d(relq) in its columns 96..DA: they go to registers, then the wave's
  // staging region (private to it: LDS operations of one wave complete in order) is reused to build whole
  // D rows -- zero-fill, 2-byte scatter through the map, 16-byte row stores.
  if (a.relD && !fold) {
    bf16_t vals[NV];
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int e = it * 64 + lane, rl = e / EXTRA, j = e % EXTRA;
      vals[it] = *(const bf16_t*)(ost + rl * OROW + (96 + j) * 2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int ldd = a.relD_ld, cpr = ldd >> 3;                   // 16-byte chunks per D row
    const int rpp = ldd <= 128 ? 32 : 8;                         // rows per pass: rpp * ldd * 2 <= 32 * OROW
    bf16_t* Dg = (bf16_t*)a.relD + ((size_t)bh * a.Nq) * ldd;
    for (int r0 = 0; r0 < 32; r0 += rpp) {
      const int n16 = rpp * cpr;
      for (int i = lane; i < n16; i += 64) *(uint4*)(ost + i * 16) = make_uint4(0u, 0u, 0u, 0u);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        const int e = it * 64 + lane, rl = e / EXTRA, j = e % EXTRA;
        if (rl >= r0 && rl < r0 + rpp && cols[it] >= 0)
          *(bf16_t*)(ost + ((rl - r0) * ldd + cols[it]) * 2) = f32_to_bf16(bf16_to_f32(vals[it]) * a.relD_scale);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for (int i = lane; i < n16; i += 64) {
        const int rl = i / cpr, ch = i % cpr;
        if (q0 + r0 + rl < a.Nq)
          *(uint4*)(Dg + (size_t)(q0 + r0 + rl) * ldd + ch * 8) = *(const uint4*)(ost + i * 16);
      }
      if (a.relX && rpp == 32) {
        // dq_extra[query, channel] = sum_k D[query, k] R^T[channel, k]: the wave's 32 D rows are in LDS, the
        // table fragments come straight from L2 (<= 24 KB, shared by every wave of the launch); swapped
        // operands as everywhere in this kernel -- the lane owns a query, its registers the channels
        f32x16_t ex[3];
#pragma unroll
        for (int pb = 0; pb < 3; ++pb)
#pragma unroll
          for (int r = 0; r < 16; ++r) ex[pb][r] = 0.f;
        const bf16_t* R = (const bf16_t*)a.relR;
        for (int ks = 0; ks < ldd / 16; ++ks) {
          const bf16x8_t dfrag = *(const bf16x8_t*)(ost + ((lane & 31) * ldd + 16 * ks + 8 * hh) * 2);
#pragma unroll
          for (int pb = 0; pb < 3; ++pb) {
            const bf16x8_t rfrag = *(const bf16x8_t*)(R + (size_t)(pb * 32 + (lane & 31)) * ldd + 16 * ks + 8 * hh);
            ex[pb] = mfma32(rfrag, dfrag, ex[pb]);
          }
        }
        if (qi < a.Nq) {
          float* xr = a.relX + ((size_t)bh * a.Nq + qi) * HD;
#pragma unroll
          for (int pb = 0; pb < 3; ++pb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *(float4*)(xr + pb * 32 + 8 * g + 4 * hh) =
                  make_float4(ex[pb][4 * g], ex[pb][4 * g + 1], ex[pb][4 * g + 2], ex[pb][4 * g + 3]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  BSTAMP(5);
#endif
}

// ---------------------------------------------------------------------------------------
// dkv kernel.  Q / dO stages of 64 queries and their (-lse2, -delta) rows arrive by LDS-DMA into
// a ring; key on the lane; dK^T / dV^T accumulate in registers over the sweep.
// NH = 1: 4 waves, each works through both 32-query halves of a stage, two-stage ring, two
// workgroups per CU.  NH = 2: 8 waves -- the four key groups twice, the halves working on the two
// 32-query halves of a stage -- so that a CU that holds ONE workgroup (the split heuristic aims at
// one per CU: every extra split costs a full fp32 atomic flush) still runs two waves per SIMD;
// three-stage ring.  The halves' dK / dV meet in LDS before the row stores.
template <int DA, int KSU, int NH>
__global__ __launch_bounds__(256 * NH, NH == 2 ? 1 : 2) void attn_bwd_dkv_kernel(svit_attn_bwd_args a,
                                                                                 int tiles_per_split) {
#if __HIP_DEVICE_COMPILE__
  constexpr int Q_BYTES = QR * DA * 2, O_BYTES = QR * HD * 2;
  constexpr int STAGE = Q_BYTES + O_BYTES + 2 * QR * 4;   // [Q | dO | -lse2 | -delta]
  constexpr int NSTAGE = NH == 2 ? 3 : 2;
  constexpr int DR = DA == 160 ? 3 : 4, DT = DA == 160 ? 2 : 3;   // read-ahead depths (register budget)
  using QLoad = BufTile<QR, DA, 4 * NH>;
  using OLoad = BufTile<QR, HD, 4 * NH>;
  constexpr int PER_STAGE = QLoad::PER_WAVE + OLoad::PER_WAVE + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int kg = wave & 3, qh = wave >> 2;                // key group, query half
#ifdef SVIT_ATTN_STAMPS
  const bool stamp_on = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0;
#endif
  BSTAMP(32);
  // all key tiles / query splits of one (batch, head) on one XCD: they stream the same Q / dO
  const int wgid = xcd_remap((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x,
                             gridDim.x * gridDim.y * gridDim.z);
  const int bx = wgid % gridDim.x, by = (wgid / gridDim.x) % gridDim.y;
  const int bh = wgid / (gridDim.x * gridDim.y), b = bh / a.heads, head = bh % a.heads;
  const int key0 = bx * 128;
  const int ki = key0 + kg * 32 + (lane & 31);
  const int kc = min(ki, a.Nk - 1);
  const bf16_t* qa = (const bf16_t*)a.qa + ((size_t)bh * a.Nq) * DA;
  const bf16_t* ka = (const bf16_t*)a.ka + ((size_t)bh * a.Nk) * DA;
  const bf16_t* vv = (const bf16_t*)a.v + ((size_t)bh * a.Nk) * HD;
  const bf16_t* dob = (const bf16_t*)a.dctx + ((size_t)b * a.Nq) * a.heads * HD + head * HD;
  const float* sc_g = a.delta + (size_t)bh * a.Nq * 2;    // [-lse2 | -delta] planes
  const int ldo = a.heads * HD;

  const int nqt = (a.Nq + QR - 1) / QR;
  const int t_begin = by * tiles_per_split;
  const int t_end = min(nqt, t_begin + tiles_per_split);
  QLoad qload;
  OLoad oload;
  qload.init(DA, wave, lane);
  oload.init(ldo, wave, lane);
  const auto qrs = __builtin_amdgcn_make_buffer_rsrc((void*)qa, 0, a.Nq * DA * 2, 0x00020000);
  const auto ors = __builtin_amdgcn_make_buffer_rsrc((void*)dob, 0, (a.Nq * ldo - head * HD) * 2, 0x00020000);
  auto issue = [&](int t) {
    unsigned char* st = smem + ((t - t_begin) % NSTAGE) * STAGE;
    const int q0 = t * QR;
    qload.issue_auto(qrs, (unsigned)q0 * DA * 2u, DA, a.Nq - q0, st, wave, lane);
    oload.issue_auto(ors, (unsigned)q0 * ldo * 2u, ldo, a.Nq - q0, st + Q_BYTES, wave, lane);
    // 64 floats of one plane = one dword LDS-DMA (rows past Nq re-read the last row): even waves
    // fetch -lse2, odd waves -delta
    const int part = wave & 1;
    const int qrow = min(q0 + lane, a.Nq - 1);
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(sc_g + (size_t)part * a.Nq + qrow),
        (__attribute__((address_space(3))) void*)(st + Q_BYTES + O_BYTES + part * 256), 4, 0, 0);
  };
  BSTAMP(33);
  if (t_begin < t_end) issue(t_begin);
  if (NSTAGE == 3 && t_begin + 1 < t_end) issue(t_begin + 1);
  BSTAMP(34);

  bf16x8_t kf[KSU], vf[6];
#pragma unroll
  for (int ks = 0; ks < KSU; ++ks) kf[ks] = *(const bf16x8_t*)(ka + (size_t)kc * DA + ks * 16 + hh * 8);
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) vf[ks] = *(const bf16x8_t*)(vv + (size_t)kc * HD + ks * 16 + hh * 8);
  // pin the register operands before the tile loop (see the dq kernel)
#pragma unroll
  for (int ks = 0; ks < KSU; ++ks) asm volatile("" : "+v"(kf[ks]));
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) asm volatile("" : "+v"(vf[ks]));

  f32x16_t dk[3], dv[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[j][r] = 0.f; dv[j][r] = 0.f; }

  BSTAMP(35);
  // NH = 2: this wave's query half is fixed -- its offset lives in the base addresses.  NH = 1: both
  // halves are unrolled and their offsets are instruction immediates.
  FragAddr fa;
  fa.init((unsigned)(size_t)smem + (NH == 2 ? qh * 2048 : 0), lane);
  const unsigned cba = (unsigned)(size_t)smem + Q_BYTES + O_BYTES + 16 * hh + (NH == 2 ? qh * 128 : 0);   // row 4*hh of the planes

  static_prio((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, wave, 4 * NH);
  for (int t = t_begin; t < t_end; ++t) {
    if (t - t_begin < 4) BSTAMP(44 + 4 * (t - t_begin));
    if (NSTAGE == 3 && t + 1 < t_end) wait_vmcnt<PER_STAGE>();   // all but the youngest stage's loads are done
    else wait_vmcnt<0>();
    if (t - t_begin < 4) BSTAMP(45 + 4 * (t - t_begin));
    __builtin_amdgcn_s_barrier();
    if (t - t_begin < 4) BSTAMP(46 + 4 * (t - t_begin));
    if (t + NSTAGE - 1 < t_end) issue(t + NSTAGE - 1);           // into the stage consumed last
    if (t - t_begin < 4) BSTAMP(47 + 4 * (t - t_begin));
    const unsigned so = ((t - t_begin) % NSTAGE) * STAGE;
    const unsigned ra[2] = {fa.rowa[0] + so, fa.rowa[1] + so};
    const unsigned ta[2] = {fa.tra[0] + so, fa.tra[1] + so};
    const unsigned ca = cba + so;
    static_for<0, (NH == 2 ? 1 : 2)>([&](auto HF) {
      constexpr int HB = NH == 2 ? 0 : decltype(HF)::value * 2048;   // 32 rows of 64 bytes per panel
      constexpr int HC = NH == 2 ? 0 : decltype(HF)::value * 128;
      const int q0 = t * QR + (NH == 2 ? qh : decltype(HF)::value) * 32;   // first query of this half's 32 rows
      // ---- initial accumulators: rows 8g + 4hh + e of the -lse2 and -delta planes -------------
      f32x4_t cl[4], cd[4];
      static_for<0, 4>([&](auto G) {
        constexpr int g = decltype(G)::value;
        lds_read128f<HC + g * 32>(cl[g], ca);
        lds_read128f<HC + 256 + g * 32>(cd[g], ca);
      });
      // ---- S (KSU k-steps of Q rows) and dP (6 k-steps of dO rows) ----------------------------
      RowStream<KSU + 6, DR> rs;
      auto rd = [&](auto J, bf16x8_t& d) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < KSU) lds_read128<HB + (j >> 1) * 4096>(d, ra[j & 1]);
        else lds_read128<HB + Q_BYTES + ((j - KSU) >> 1) * 4096>(d, ra[(j - KSU) & 1]);
      };
      rs.prologue(rd);
      // the DR prologue reads are younger than the eight constant reads: lgkmcnt(DR) = constants landed
      asm volatile("s_waitcnt lgkmcnt(%8)"
                   : "+v"(cl[0]), "+v"(cl[1]), "+v"(cl[2]), "+v"(cl[3]), "+v"(cd[0]), "+v"(cd[1]),
                     "+v"(cd[2]), "+v"(cd[3]) : "n"(DR) : "memory");
      f32x16_t s, dp;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[4 * g + e] = cl[g][e]; dp[4 * g + e] = cd[g][e]; }
      rs.run(rd, [&](auto J, const bf16x8_t& f) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < KSU) s = mfma32(f, kf[j], s);
        else dp = mfma32(f, vf[j - KSU], dp);
      });
      // ---- dO^T / Q^T fragments of the dV / dK products: the first travel under the VALU section
      TrStream<12, DT> ts;
      auto rdt = [&](auto J, s16x4_t& lo, s16x4_t& hi) {
        constexpr int j = decltype(J)::value, sp = j / 6, w = (j % 6) / 3, p = j % 3;
        lds_read_tr<HB + (w == 0 ? Q_BYTES : 0) + sp * 16 * 64 + p * 4096>(lo, ta[0]);
        lds_read_tr<HB + (w == 0 ? Q_BYTES : 0) + sp * 16 * 64 + p * 4096>(hi, ta[1]);
      };
      ts.prologue(rdt);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r]);
      if (q0 + 32 > a.Nq) {      // ragged last stage (uniform branch): re-read rows past Nq count nothing
        asm volatile("; ragged query rows" ::: "memory");   // keeps this a branch (no per-element selects)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (q0 + acc_row(r, lane) >= a.Nq) s[r] = 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = s[r] * dp[r];
      const bf16x8_t pf0 = acc_to_frag(s, 0), pf1 = acc_to_frag(s, 1);
      const bf16x8_t df0 = acc_to_frag(dp, 0), df1 = acc_to_frag(dp, 1);
      ts.run(rdt, [&](auto J, const bf16x8_t& f) {
        constexpr int j = decltype(J)::value, sp = j / 6, w = (j % 6) / 3, p = j % 3;
        if constexpr (w == 0) dv[p] = mfma32(f, sp == 0 ? pf0 : pf1, dv[p]);
        else dk[p] = mfma32(f, sp == 0 ? df0 : df1, dk[p]);
      });
    });
  }
  BSTAMP(36);
  // ---- epilogue: each lane holds, for its key row, four 16-byte runs of every 32-column block
  // (columns 32j + 8g + 4hh + 0..3): they go to this block's dk / dv plane as they stand.  With two
  // query halves the second half first hands its accumulators to the first through LDS, in
  // register layout ([run][thread] float4: conflict-free both ways).
  if (NH == 2) {
    __syncthreads();                       // every wave is done with the ring
    float4* xb = (float4*)smem;            // [2 tensors][12 runs][256 threads]
    const int th = kg * 64 + lane;
    if (qh == 1) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          xb[(j * 4 + g) * 256 + th] = make_float4(dk[j][4 * g], dk[j][4 * g + 1], dk[j][4 * g + 2], dk[j][4 * g + 3]);
          xb[(12 + j * 4 + g) * 256 + th] = make_float4(dv[j][4 * g], dv[j][4 * g + 1], dv[j][4 * g + 2], dv[j][4 * g + 3]);
        }
    }
    __syncthreads();
    if (qh == 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 x = xb[(j * 4 + g) * 256 + th], y = xb[(12 + j * 4 + g) * 256 + th];
          dk[j][4 * g] += x.x; dk[j][4 * g + 1] += x.y; dk[j][4 * g + 2] += x.z; dk[j][4 * g + 3] += x.w;
          dv[j][4 * g] += y.x; dv[j][4 * g + 1] += y.y; dv[j][4 * g + 2] += y.z; dv[j][4 * g + 3] += y.w;
        }
    }
  }
  BSTAMP(37);
  if (qh == 0 && ki < a.Nk) {
    const size_t plane = (size_t)a.B * a.heads * a.Nk * HD;
    const size_t row = ((size_t)by * a.B * a.heads + bh) * a.Nk + ki;
    float* dkp = a.dk + row * HD + 4 * hh;
    float* dvp = a.dv + row * HD + 4 * hh;
    (void)plane;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *(float4*)(dkp + j * 32 + 8 * g) = make_float4(dk[j][4 * g] * a.scale, dk[j][4 * g + 1] * a.scale,
                                                       dk[j][4 * g + 2] * a.scale, dk[j][4 * g + 3] * a.scale);
        *(float4*)(dvp + j * 32 + 8 * g) = make_float4(dv[j][4 * g], dv[j][4 * g + 1], dv[j][4 * g + 2], dv[j][4 * g + 3]);
      }
  }
  BSTAMP(38);
#endif
}

#ifndef SVIT_DKV_TARGET       // workgroups the query split of the dkv kernel aims at (swept inside the step in round 4)
#define SVIT_DKV_TARGET 256
#endif
constexpr int g_dkv_target = SVIT_DKV_TARGET;
// (timing-only "run one of the two kernels" switch of the diagnostic builds, -DSVIT_DIAG_BWD_ONLY=1 dq / 2 dkv: the skipped
// outputs are NOT written, so it cannot exist in the product library -- tools/diag/build_variant.py)
#ifndef SVIT_DIAG_BWD_ONLY
#define SVIT_DIAG_BWD_ONLY 0
#endif

// how the query range of the dkv kernel is cut: (dkv waves / 4, effective number of parts)
struct DkvPlan { int halves, splits, tiles_per_split; };
static DkvPlan dkv_plan(const svit_attn_bwd_args& a) {
  const int key_blocks = (a.Nk + 127) / 128;
  const int base = key_blocks * a.B * a.heads;
  // two query halves (8 waves) where the launch leaves one 4-wave workgroup per CU anyway: the
  // short-key blocks (tools/bench_kernels.py attn)
  int halves = svit_knob(SVIT_K_ATTN_DKV_FORM);
  if (halves != 1 && halves != 2) halves = (a.DA == 128 && base <= 256) ? 2 : 1;
  const int nqt = (a.Nq + QR - 1) / QR;
  int splits = a.q_splits;
  if (splits <= 0) {
    // split the query range only as far as needed to fill the chip (~1 block per CU: every part is
    // one more dk / dv plane for the consumer to read) and keep >= 2 query stages per block
    splits = (g_dkv_target + base - 1) / base;
    if (splits > nqt / 2) splits = nqt / 2;
  }
  if (splits > nqt) splits = nqt;
  if (splits < 1) splits = 1;
  const int tiles_per_split = (nqt + splits - 1) / splits;
  splits = (nqt + tiles_per_split - 1) / tiles_per_split;
  return {halves, splits, tiles_per_split};
}

template <int DA, int KSU>
int launch_bwd(const svit_attn_bwd_args& a, hipStream_t st) {
  static SvitOnce once_dq, once_kv, once_kv2;
  size_t lds_dq = 2 * (size_t)(KT * DA * 2 + KT * HD * 2);
  const size_t lds_dq_out = (size_t)4 * 32 * (DA * 2 + 16) + (size_t)4 * 32 * 128 * 2;   // + the fold mode's D rows
  if (lds_dq < lds_dq_out) lds_dq = lds_dq_out;
  // the Q staging region behind stage 0 (attn_common.h RowStage: 8 / 11 KiB per wave)
  lds_dq = std::max(lds_dq, (size_t)(KT * DA * 2 + KT * HD * 2) + 4 * (size_t)attn::RowStage<DA>::BYTES);
  const size_t stage = (size_t)(QR * DA * 2 + QR * HD * 2 + 2 * QR * 4);
  size_t lds_kv = 2 * stage, lds_kv2 = 3 * stage;
  const size_t lds_merge = (size_t)2 * 12 * 256 * 16;     // the halves' dk / dv hand-over
  if (lds_kv2 < lds_merge) lds_kv2 = lds_merge;
  if (int rc = svit_max_lds_once(once_dq, (const void*)attn_bwd_dq_kernel<DA, KSU>, lds_dq)) return rc;
  if (int rc = svit_max_lds_once(once_kv, (const void*)attn_bwd_dkv_kernel<DA, KSU, 1>, lds_kv)) return rc;
  if (int rc = svit_max_lds_once(once_kv2, (const void*)attn_bwd_dkv_kernel<DA, KSU, 2>, lds_kv2)) return rc;
  const DkvPlan pl = dkv_plan(a);
  const int key_blocks = (a.Nk + 127) / 128;
  if (((uintptr_t)a.dk | (uintptr_t)a.dv) & 15) return SVIT_ERR_ALIGN;
  constexpr int skip = SVIT_DIAG_BWD_ONLY == 1 ? 2 : SVIT_DIAG_BWD_ONLY == 2 ? 1 : 0;
  if (!(skip & 1))
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DA, KSU>), dim3((a.Nq + 127) / 128, a.B * a.heads), dim3(256),
                       lds_dq, st, a);
  SVIT_LAUNCH_CHECK();
  if (skip & 2) return SVIT_OK;
  if (pl.halves == 2)
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DA, KSU, 2>), dim3(key_blocks, pl.splits, a.B * a.heads), dim3(512),
                       lds_kv2, st, a, pl.tiles_per_split);
  else
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DA, KSU, 1>), dim3(key_blocks, pl.splits, a.B * a.heads), dim3(256),
                       lds_kv, st, a, pl.tiles_per_split);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}
}  // namespace

static int check_bwd_args(const svit_attn_bwd_args* a) {
  if (!a) return SVIT_ERR_ARG;
  if (a->B <= 0 || a->heads <= 0 || a->Nq <= 0 || a->Nk <= 0) return SVIT_ERR_SHAPE;
  if (a->B * a->heads > 65535) return SVIT_ERR_SHAPE;
  if (a->DA != 128 && a->DA != 160) return SVIT_ERR_SHAPE;
  if (a->bias_cols < 0 || a->bias_cols > a->DA - 96) return SVIT_ERR_ARG;
  return SVIT_OK;
}

extern "C" int svit_attn_bwd_parts(const svit_attn_bwd_args* a) {
  if (int rc = check_bwd_args(a)) return rc;
  return dkv_plan(*a).splits;
}

extern "C" int svit_attn_bwd(const svit_attn_bwd_args* a, void* stream) {
  if (!a || !a->qa || !a->ka || !a->v || !a->ctx || !a->dctx || !a->lse2 || !a->delta || !a->dqa ||
      !a->dk || !a->dv)
    return SVIT_ERR_ARG;
  if (a->B <= 0 || a->heads <= 0 || a->Nq <= 0 || a->Nk <= 0) return SVIT_ERR_SHAPE;
  if (a->B * a->heads > 65535) return SVIT_ERR_SHAPE;
  if (a->DA != 128 && a->DA != 160) return SVIT_ERR_SHAPE;
  if (a->bias_cols < 0 || a->bias_cols > a->DA - 96) return SVIT_ERR_ARG;
  if (((uintptr_t)a->qa | (uintptr_t)a->ka | (uintptr_t)a->v | (uintptr_t)a->dctx | (uintptr_t)a->dqa) & 15)
    return SVIT_ERR_ALIGN;
  if (a->relD) {
    if (!a->relD_map || a->relD_ld <= 0 || a->relD_ld % 8 != 0 || a->relD_ld > 544) return SVIT_ERR_ARG;
    if ((uintptr_t)a->relD & 15) return SVIT_ERR_ALIGN;
    if (a->relX && (!a->relR || a->relD_ld > 128 || a->relD_ld % 16 != 0 ||
                    (((uintptr_t)a->relR | (uintptr_t)a->relX) & 15)))
      return SVIT_ERR_ARG;
    if (a->relR && !a->relX && (a->relD_ld > 128 || a->relD_ld % 16 != 0 || ((uintptr_t)a->relR & 15) ||
                                ((uintptr_t)a->relD_map & 15)))
      return SVIT_ERR_ARG;     // fold mode exists for narrow tables only
  } else if (a->relX || a->relR) {
    return SVIT_ERR_ARG;
  }
  const int bias_cols = a->bias_cols > 0 ? a->bias_cols : a->DA - 96;
  const int ksu = 6 + (bias_cols + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  if (a->DA == 128) return ksu <= 7 ? launch_bwd<128, 7>(*a, st) : launch_bwd<128, 8>(*a, st);
  return ksu <= 9 ? launch_bwd<160, 9>(*a, st) : launch_bwd<160, 10>(*a, st);
}
