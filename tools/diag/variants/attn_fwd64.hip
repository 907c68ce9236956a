// Fused pooled attention, forward -- the one-wave-per-SIMD, 64-query-rows-per-wave form (round 4) -- gfx950.
//
// Same operands, same arithmetic conventions and the same LDS machinery as attn_fwd.hip (log2-domain scores with the
// rel-pos bias inside the QK^T contraction, the running maximum as the initial accumulator, BufTile LDS-DMA of 64-key
// K / V tiles, RowStream / TrStream fragment streams, RowStage Q fetch).  What differs is the shape of the loop
// (the structure of the guide's "4-wave, one-wave-per-SIMD, persistent" kernel):
//   * a workgroup = 4 waves = 256 query rows, launch_bounds(256, 1): ONE wave per SIMD with the whole register file; a
//     wave owns TWO 32-query blocks A and B (64 rows), so a K / V tile is fetched once per 256 queries (half the LDS-DMA
//     issue per flop of the 128-query form) -- one workgroup per CU;
//   * the two blocks are each other's cover: inside ONE instruction stream the matrix pipe runs block A's products
//     while the vector pipe finishes block B's softmax and vice versa, slot by slot --
//         slot 1: MFMA  S(A,t) = K(t) Qa^T          | VALU  second half of exp + bf16 packing of P(B,t-1)
//         slot 2: MFMA  O(B) += V(t-1)^T P(B,t-1)   | VALU  row max / re-base test of S(A,t), first half of its exp
//         slot 3: MFMA  S(B,t) = K(t) Qb^T          | VALU  second half of exp + packing of P(A,t)
//         slot 4: MFMA  O(A) += V(t)^T P(A,t)       | VALU  row max / re-base test of S(B,t), first half of its exp
//     every MFMA is followed by <= 5 vector instructions of the other block and a sched_barrier, so the order survives;
//   * row sums of P are fp32 adds in those gaps (the 128-query kernel spends 4 MFMAs per tile and wave on them);
//   * a three-stage K / V ring: block B's P.V of tile t-1 runs during tile t, so V(t-1) stays while K(t) is consumed
//     and tile t+1 lands.
// Selected by svit_attn_fwd's heuristic / SVIT_ATTN_FWD_W64 (attn_fwd.hip); measured in profiles/r04_attn_w64.txt.
#include <algorithm>
#include <cstdlib>
#include "attn_common.h"
#include "../../include/svit_hip.h"

namespace {
using namespace attn;
constexpr int KT = 64;  // keys per tile

__device__ __forceinline__ float other_half64(float x) {      // lane l <-> lane l ^ 32, VALU only
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

// one 32-query block's softmax state.  At one wave per SIMD hipcc selects the AGPR form for every MFMA: the score
// accumulators (acc0, acc1) and their initial value (negm) live in the accumulator half of the register file and are
// written by MFMAs only (negm: on a re-base).  The vector pipe works on ONE explicit copy of the scores, p[32], taken
// eight values per MFMA gap while the row maximum is folded (v_accvgpr_read is the copy) and never written back.
struct QBlock {
  f32x16_t acc0, acc1;   // S^T of the tile's two 32-key blocks (lane = query, registers = keys): MFMA-side
  float p[32];           // the scores / probabilities on the vector side
  bf16x8_t pf[4];        // P as the B operand of the P.V MFMAs: key groups 0..3 of 16
  float m_run, lsum, mx;
};

constexpr float RESCALE_THR = 6.0f;

// ---- V1: copy + row maximum, re-base test, exponentials of the first key block; NM = MFMA steps of its slot -----------
// steps 0..3: 8 scores per step copied out of the accumulators and folded into mx; step 4: other half, test, (rare)
// re-base; steps 5..NM-1: the 16 exponentials of the first key block with their row-sum adds
template <int J, int NM>
__device__ __forceinline__ void v1_step(QBlock& x, f32x16_t (&o)[3], bool first, int kvalid) {
  if constexpr (J < 4) {
    const f32x16_t& s = J < 2 ? x.acc0 : x.acc1;
    constexpr int b = (J & 1) * 8, pb = J * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) x.p[pb + e] = s[b + e];
    if (kvalid < KT) {        // ragged last tile (uniform): key rows past Nk hold re-read data
      const int lane = threadIdx.x & 63;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if ((J >> 1) * 32 + acc_row(b + e, lane) >= kvalid) x.p[pb + e] = -INFINITY;
    }
    if constexpr (J == 0) {
      x.mx = max3(x.p[0], x.p[1], x.p[2]);
      x.mx = max3(x.mx, x.p[3], x.p[4]);
      x.mx = max3(x.mx, x.p[5], x.p[6]);
      x.mx = fmaxf(x.mx, x.p[7]);
    } else {
      x.mx = max3(x.mx, x.p[pb], x.p[pb + 1]);
      x.mx = max3(x.mx, x.p[pb + 2], x.p[pb + 3]);
      x.mx = max3(x.mx, x.p[pb + 4], x.p[pb + 5]);
      x.mx = max3(x.mx, x.p[pb + 6], x.p[pb + 7]);
    }
  } else if constexpr (J == 4) {
    x.mx = fmaxf(x.mx, other_half64(x.mx));
    // defer-max (attn_fwd.hip): re-base only when the maximum grew by more than 2^THR; the first tile always does
    if (first || !__all(x.mx <= RESCALE_THR)) {
      asm volatile("; re-base" ::: "memory");
      const float shift = first ? x.mx : fmaxf(x.mx, 0.f);
      if (!first) {
        const float alpha = fast_exp2(-shift);
        x.lsum *= alpha;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[j][r] *= alpha;
      }
      x.m_run += shift;
#pragma unroll
      for (int r = 0; r < 32; ++r) x.p[r] -= shift;
    }
  } else {
    constexpr int NE = NM - 5;                       // steps left for 16 exponentials
    constexpr int lo = (J - 5) * 16 / NE, hi = (J - 4) * 16 / NE;
#pragma unroll
    for (int r = lo; r < hi; ++r) {
      x.p[r] = fast_exp2(x.p[r]);
      x.lsum += x.p[r];
    }
  }
}
// ---- V2: exponentials of the second key block and the bf16 packing of all four P fragments, over NM >= 16 steps -------
template <int J, int NM>
__device__ __forceinline__ void v2_step(QBlock& x) {
  if constexpr (J < 16) {
    x.p[16 + J] = fast_exp2(x.p[16 + J]);
    x.lsum += x.p[16 + J];
    // packing pair J: fragments 0, 1 from the first key block (exponentiated in V1), 2, 3 from the second (pair J needs
    // p[16 + 2 (J - 8) + 1] <= p[16 + J])
    constexpr int k = J >> 2, e = (J & 3) * 2;      // fragment k, elements e, e + 1
    constexpr int base = (k >> 1) * 16 + (k & 1) * 8;
    x.pf[k][e] = (__bf16)x.p[base + e];
    x.pf[k][e + 1] = (__bf16)x.p[base + e + 1];
  }
}

template <int KSU>
__global__ __launch_bounds__(256, 1) void attn_fwd_w64_kernel(svit_attn_fwd_args a) {
#if __HIP_DEVICE_COMPILE__
  constexpr int NS = 3;
  constexpr int NP = (KSU + 1) / 2, KCOLS = NP * 32;
  constexpr int K_BYTES = KT * KCOLS * 2, V_BYTES = KT * HD * 2, STAGE = K_BYTES + V_BYTES;
  using KLoad = BufTile<KT, KCOLS, 4>;
  using VLoad = BufTile<KT, HD, 4>;
  using QStage = RowStage<(KSU <= 8 ? 128 : 160)>;
  constexpr int NQK = 2 * KSU;                       // MFMAs of a QK^T slot (two 32-key blocks)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int DA = a.DA;
  const int wgid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int bh = wgid / gridDim.x, b = bh / a.heads, head = bh % a.heads;
  const int q0 = (wgid % gridDim.x) * 256 + wave * 64;
  const bf16_t* qa = (const bf16_t*)a.qa + ((size_t)bh * a.Nq) * DA;
  const bf16_t* ka = (const bf16_t*)a.ka + ((size_t)bh * a.Nk) * DA;
  const bf16_t* vv = (const bf16_t*)a.v + ((size_t)bh * a.Nk) * HD;

  f32x16_t o[2][3];
  QBlock blk[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[x][j][r] = 0.f;
    blk[x].m_run = 0.f; blk[x].lsum = 0.f; blk[x].mx = 0.f;
  }

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

  const int nt = (a.Nk + KT - 1) / KT;
  KLoad kload;
  VLoad vload;
  kload.init(DA, wave, lane);
  vload.init(HD, wave, lane);
  const auto krs = __builtin_amdgcn_make_buffer_rsrc((void*)ka, 0, a.Nk * DA * 2, 0x00020000);
  const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vv, 0, a.Nk * HD * 2, 0x00020000);
  auto issue = [&](int t) {
    unsigned char* st = smem + (t % NS) * STAGE;
    const unsigned k0 = (unsigned)t * KT;
    kload.issue_auto(krs, k0 * DA * 2u, DA, a.Nk - (int)k0, st, wave, lane);
    vload.issue_auto(vrs, k0 * HD * 2u, HD, a.Nk - (int)k0, st + K_BYTES, wave, lane);
  };
  issue(0);
  // Q rows of both blocks through LDS, coalesced, into the stages the prologue leaves free
  bf16x8_t qf[2][KSU];
  {
    unsigned char* qreg = smem + STAGE + wave * (2 * QStage::BYTES);
    QStage::issue(qa, DA, q0, a.Nq, qreg, lane);
    QStage::issue(qa, DA, q0 + 32, a.Nq, qreg + QStage::BYTES, lane);
    wait_vmcnt<0>();
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int ks = 0; ks < KSU; ++ks) qf[x][ks] = QStage::frag(qreg + x * QStage::BYTES, ks, lane);
  }
  // pin the Q fragments before the loop (their loads must not sink into it) -- in ACCUMULATOR registers: they are
  // only ever MFMA B operands, and 64-80 architectural registers is what the softmax state of two blocks needs
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int ks = 0; ks < KSU; ++ks) asm volatile("" : "+a"(qf[x][ks]));

  // QK^T slot of block X on tile t's K image, with the V2 steps of block Y (tile of the slot before) in its gaps
  auto qk_slot = [&](auto X, unsigned so, auto WithV2, auto Y) {
    constexpr bool V2 = decltype(WithV2)::value == 1;
    constexpr int x = decltype(X)::value, y = decltype(Y)::value;
    const unsigned kaddr[2] = {kaddr0[0] + so, kaddr0[1] + so};
    // the initial accumulator of both chains: -running maximum in all 16 registers, built per slot (one register block
    // shared by the two query blocks: with two resident copies the accumulator half of the file over-subscribes at
    // KSU >= 9 -- 80 Q + 96 O + 64 S + 32)
    f32x16_t negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = -blk[x].m_run;
    RowStream<NQK, 4> ks_;
    auto rdk = [&](auto J, bf16x8_t& d) {
      constexpr int j = decltype(J)::value, kb = j / KSU, ks = j % KSU;
      lds_read128<kb * 2048 + (ks >> 1) * KT * 64>(d, kaddr[ks & 1]);
    };
    ks_.prologue(rdk);
    ks_.run(rdk, [&](auto J, const bf16x8_t& f) {
      constexpr int j = decltype(J)::value, kb = j / KSU, ks = j % KSU;
      if constexpr (kb == 0) blk[x].acc0 = mfma32(f, qf[x][ks], ks == 0 ? negm : blk[x].acc0);
      else blk[x].acc1 = mfma32(f, qf[x][ks], ks == 0 ? negm : blk[x].acc1);
      if constexpr (V2) v2_step<j, NQK>(blk[y]);
    });
    if constexpr (V2 && NQK < 16)       // (KSU = 7: two steps of V2 have no MFMA to hide behind)
      static_for<NQK, 16>([&](auto J) { v2_step<decltype(J)::value, 16>(blk[y]); });
  };
  // P.V slot of block X on the V image at stage offset so, with the V1 steps of block Y in its gaps
  auto pv_slot = [&](auto X, unsigned so, auto WithV1, auto Y, bool first, int kvalid) {
    constexpr bool V1 = decltype(WithV1)::value == 1;
    constexpr int x = decltype(X)::value, y = decltype(Y)::value;
    const unsigned vaddr[2] = {vaddr0[0] + so, vaddr0[1] + so};
    TrStream<12, 3> vs;
    auto rdv = [&](auto J, s16x4_t& lo, s16x4_t& hi) {
      constexpr int i = decltype(J)::value, g = i / 3, j = i % 3;
      lds_read_tr<g * 16 * 64 + j * KT * 64>(lo, vaddr[0]);
      lds_read_tr<g * 16 * 64 + j * KT * 64>(hi, vaddr[1]);
    };
    vs.prologue(rdv);
    vs.run(rdv, [&](auto J, const bf16x8_t& f) {
      constexpr int i = decltype(J)::value, g = i / 3, j = i % 3;
      o[x][j] = mfma32(f, blk[x].pf[g], o[x][j]);
      if constexpr (V1) v1_step<i, 12>(blk[y], o[y], first, kvalid);
    });
  };
  auto v1_alone = [&](auto Y, bool first, int kvalid) {
    constexpr int y = decltype(Y)::value;
    static_for<0, 12>([&](auto J) { v1_step<decltype(J)::value, 12>(blk[y], o[y], first, kvalid); });
  };
  auto v2_alone = [&](auto Y) {
    constexpr int y = decltype(Y)::value;
    static_for<0, 16>([&](auto J) { v2_step<decltype(J)::value, 16>(blk[y]); });
  };

  for (int t = 0; t < nt; ++t) {
    wait_vmcnt<0>();                 // this wave's share of tile t has landed (tile t + 1 is issued below)
    __builtin_amdgcn_s_barrier();    // everyone's has; everyone is past slot 2 of tile t - 1 (the last reader of V(t-2))
    if (t + 1 < nt) issue(t + 1);    // into the stage tile t - 2 left
    const unsigned so = (t % NS) * STAGE, sp = ((t + NS - 1) % NS) * STAGE;
    const int kvalid = a.Nk - t * KT;                  // >= KT except on a ragged last tile
    if (t == 0) {
      qk_slot(Int<0>{}, so, Int<0>{}, Int<1>{});                       // S(A,0)
      v1_alone(Int<0>{}, true, kvalid);
      qk_slot(Int<1>{}, so, Int<1>{}, Int<0>{});                       // S(B,0)     | V2(A,0)
      pv_slot(Int<0>{}, so, Int<1>{}, Int<1>{}, true, kvalid);         // O(A) += .. | V1(B,0)
    } else {
      qk_slot(Int<0>{}, so, Int<1>{}, Int<1>{});                       // S(A,t)     | V2(B,t-1)
      pv_slot(Int<1>{}, sp, Int<1>{}, Int<0>{}, false, kvalid);        // O(B) += V(t-1)^T P(B,t-1) | V1(A,t)
      qk_slot(Int<1>{}, so, Int<1>{}, Int<0>{});                       // S(B,t)     | V2(A,t)
      pv_slot(Int<0>{}, so, Int<1>{}, Int<1>{}, false, kvalid);        // O(A) += V(t)^T P(A,t)     | V1(B,t)
    }
  }
  // drain: block B's last tile
  v2_alone(Int<1>{});
  pv_slot(Int<1>{}, ((nt - 1) % NS) * STAGE, Int<0>{}, Int<0>{}, false, KT);

  // ---- epilogue (per block, as attn_fwd.hip): normalise, stage 32 x 96 in LDS, add the pooled q, store whole rows ----
  __builtin_amdgcn_s_barrier();        // every wave is done with the K/V ring
  constexpr int OROW = 208;
  unsigned char* ost = smem + wave * (32 * OROW);
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int qb = q0 + 32 * x, qi = qb + (lane & 31);
    uint4 qres[6];
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int id = it * 64 + lane, row = id / 12, ch = id % 12;
      const int q = min(qb + row, a.Nq - 1);
      qres[it] = *(const uint4*)(qa + (size_t)q * DA + ch * 8);
    }
    const float l = blk[x].lsum + other_half64(blk[x].lsum);
    const float inv = 1.f / l;
    if (hh == 0 && qi < a.Nq) a.lse2[(size_t)bh * a.Nq + qi] = blk[x].m_run + log2f(l);
    {
      unsigned char* orow = ost + (lane & 31) * OROW;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dv = j * 32 + 8 * g + 4 * hh;
          uint2 pk;
          pk.x = pack_bf16x2(o[x][j][4 * g] * inv, o[x][j][4 * g + 1] * inv);
          pk.y = pack_bf16x2(o[x][j][4 * g + 2] * inv, o[x][j][4 * g + 3] * inv);
          *(uint2*)(orow + dv * 2) = pk;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int id = it * 64 + lane, row = id / 12, ch = id % 12;
      const int q = qb + row;
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
#endif
}

template <int KSU>
int launch_w64(const svit_attn_fwd_args& a, hipStream_t st) {
  constexpr int NP = (KSU + 1) / 2;
  const size_t stage = (size_t)(KT * NP * 32 * 2 + KT * HD * 2);
  size_t lds = 3 * stage;
  lds = std::max(lds, stage + 8 * (size_t)attn::RowStage<(KSU <= 8 ? 128 : 160)>::BYTES);
  lds = std::max(lds, (size_t)4 * 32 * 208);
  static SvitOnce once;
  if (int rc = svit_max_lds_once(once, (const void*)attn_fwd_w64_kernel<KSU>, lds)) return rc;
  dim3 grid((a.Nq + 255) / 256, a.B * a.heads);
  hipLaunchKernelGGL((attn_fwd_w64_kernel<KSU>), grid, dim3(256), lds, st, a);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}
}  // namespace

// called by svit_attn_fwd (attn_fwd.hip) after its argument checks; ksu = 6 + ceil(bias columns / 16)
int attn_fwd_w64_launch(const svit_attn_fwd_args* a, int ksu, void* stream) {
  switch (ksu) {
    case 7: return launch_w64<7>(*a, (hipStream_t)stream);
    case 8: return launch_w64<8>(*a, (hipStream_t)stream);
    case 9: return launch_w64<9>(*a, (hipStream_t)stream);
    case 10: return launch_w64<10>(*a, (hipStream_t)stream);
  }
  return SVIT_ERR_SHAPE;
}
