// Fused pooled attention, forward (attention.py:429-461) -- gfx950.
//
//   S = (q*scale) k^T + rel-pos bias ; P = softmax(S) ; ctx = P v + q (all tokens but cls)
//
// The decomposed relative-position bias rides inside the QK^T MFMA: the query operand is
// augmented to qa = [q | relq/scale] and the key operand to ka = [k | one-hot(y,x,t)], so
// qa.ka^T * scale = scale*q.k + rel_h[y] + rel_w[x] + rel_t[t]; cls/object rows/cols carry
// zeros there.  Head dim of the contraction DA = 128 or 160, value dim 96.
//
// Work split: block = 4 waves x 32 queries; K/V tiles of 64 keys arrive by LDS-DMA
// (global_load_lds) in a two-stage, swizzled LDS panel image (attn_common.h), one raw barrier
// per tile behind the wave's own vmcnt wait.  Scores are
// computed "swapped" (S^T = ka qa^T): every lane owns ONE query column and 16 key rows per
// 32x32 block, so the online-softmax row reduction is in-register plus one lane<->lane+32
// exchange, and the exponentiated tile feeds the PV MFMA as its B operand without leaving
// registers.  O^T accumulates as 3 x (32 dv x 32 query) blocks.
#include <cstdlib>
#include <type_traits>
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

namespace {
using namespace attn;
constexpr int KT = 64;  // keys per tile

// ---- explicitly pipelined LDS fragment reads -------------------------------------------------
// Left to itself hipcc emits `ds_read_b128 ; s_waitcnt lgkmcnt(0) ; v_mfma` per k-step -- every
// MFMA waits out the full LDS latency of its own operand (measured: the QK^T phase ran at a
// third of the MFMA rate).  The reads are therefore issued by hand, one chunk of G fragments
// AHEAD of the MFMAs that consume the previous chunk, through inline asm (invisible to the
// waitcnt pass), and released by a counted `s_waitcnt lgkmcnt(N)` that carries the fragment
// registers as in/out operands so that no consumer can be scheduled above it.  LDS operations
// return in order, so "at most N outstanding" = everything older than the last N has landed.
template <int I> using Int = std::integral_constant<int, I>;
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(Int<B>{});
    static_for<B + 1, E>(f);
  }
}
template <int OFF>
__device__ __forceinline__ void lds_read128(bf16x8_t& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "i"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read_tr(s16x4_t& d, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_release(bf16x8_t (&f)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_release(bf16x8_t (&f)[5]) {
  asm volatile("s_waitcnt lgkmcnt(%5)"
               : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_release(s16x4_t (&f)[6]) {
  asm volatile("s_waitcnt lgkmcnt(%6)"
               : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]) : "n"(N) : "memory");
}

// NW waves per workgroup share one K/V stream (NW*32 queries); NS = depth of the K/V ring.
// <4, 2>: two independent workgroups per CU.  <8, 3>: one 8-wave workgroup per CU streams each
// K/V tile ONCE for 256 queries and prefetches two tiles ahead.  Cycle stamps of the <4, 2> loop
// (tools/attn_stamps.py, 6337 x 1633, DA 160): ~4100 cycles per tile = DMA issue 1150 (a wave
// is held ~140 cycles per 1-KiB piece, wherever in the tile the piece is placed) + QK^T 1030 +
// softmax 380 + PV 1000 + wait/barrier 450, data landing ~3900 cycles after its first piece.
template <int DA, int NW, int NS>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void attn_fwd_kernel(svit_attn_fwd_args a) {
  constexpr int KS = DA / 16;                 // k-steps of the QK^T contraction
  constexpr int K_BYTES = KT * DA * 2, V_BYTES = KT * HD * 2;
  constexpr int STAGE = K_BYTES + V_BYTES;    // [K tile | V tile] per pipeline stage
  using KLoad = GldsTile<KT, DA, NW>;
  using VLoad = GldsTile<KT, HD, NW>;
  constexpr int PIECES = KLoad::PER_WAVE + VLoad::PER_WAVE;   // DMA instructions per wave and tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int wgid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int bh = wgid / gridDim.x, b = bh / a.heads, head = bh % a.heads;
  const int q0 = (wgid % gridDim.x) * (NW * 32) + wave * 32;
  const int qi = q0 + (lane & 31);
  const int qc = min(qi, a.Nq - 1);
  const bf16_t* qa = (const bf16_t*)a.qa + ((size_t)bh * a.Nq) * DA;
  const bf16_t* ka = (const bf16_t*)a.ka + ((size_t)bh * a.Nk) * DA;
  const bf16_t* vv = (const bf16_t*)a.v + ((size_t)bh * a.Nk) * HD;
  const float c = a.scale * 1.4426950408889634f;

  bf16x8_t qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    qf[ks] = *(const bf16x8_t*)(qa + (size_t)qc * DA + ks * 16 + hh * 8);
  // pin the register operands before the tile loop: their first use must not sit inside it,
  // or the compiler's wait for them (vmcnt(0)) would drain the LDS-DMA pipeline every tile
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));

  f32x16_t o[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
  // running max kept in the exp2 domain (already multiplied by c); l = partial row sum
  float m_run = -INFINITY, l_run = 0.f;

  // per-lane LDS byte addresses of the fragment reads (stage 0; everything else is an
  // immediate): K row fragments of k-step ks sit at kaddr[ks&1] + (ks>>1)*KT*64 + kb*2048, the
  // transposed V fragments of key group rbase / panel j at vaddr[0|1] + rbase*64 + j*KT*64
  // (image and swizzle: attn_common.h).
  constexpr int G = DA == 128 ? 4 : 5;          // fragments per chunk; 2*KS/G chunks per tile
  constexpr int NC = 2 * KS / G;
  static_assert(NC * G == 2 * KS, "chunking");
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
  auto issue = [&](int t) {
    unsigned char* st = smem + (t % NS) * STAGE;
    const int k0 = t * KT;
    kload.issue_auto(ka + (size_t)k0 * DA, DA, a.Nk - k0, st, wave, lane);
    vload.issue_auto(vv + (size_t)k0 * HD, HD, a.Nk - k0, st + K_BYTES, wave, lane);
  };
  issue(0);
  if (NS == 3 && nt > 1) issue(1);
  for (int t = 0; t < nt; ++t) {
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
    // V^T fragments of key group g (16 keys: kb = g>>1, sp = g&1), 3 panels x (lo, hi)
    s16x4_t vt[2][6];
    auto issue_v = [&](auto Gi, s16x4_t (&d)[6]) {
      constexpr int g = decltype(Gi)::value;
      static_for<0, 3>([&](auto J) {
        constexpr int j = decltype(J)::value;
        lds_read_tr<g * 16 * 64 + j * KT * 64>(d[2 * j], vaddr[0]);
        lds_read_tr<g * 16 * 64 + j * KT * 64>(d[2 * j + 1], vaddr[1]);
      });
    };
    issue_v(Int<0>{}, vt[0]);        // lands under the whole QK^T + softmax phase
    // ---- S^T = ka . qa^T for the two 32-key blocks of the tile --------------------------
    f32x16_t s[2];
    bf16x8_t kf[2][G];
    auto issue_k = [&](auto Ci, bf16x8_t (&d)[G]) {
      constexpr int c = decltype(Ci)::value;
      static_for<0, G>([&](auto J) {
        constexpr int j = c * G + decltype(J)::value, kb = j / KS, ks = j % KS;
        lds_read128<kb * 2048 + (ks >> 1) * KT * 64>(d[decltype(J)::value], kaddr[ks & 1]);
      });
    };
    issue_k(Int<0>{}, kf[0]);
    static_for<0, NC>([&](auto Ci) {
      constexpr int c = decltype(Ci)::value;
      if constexpr (c + 1 < NC) {
        issue_k(Int<c + 1>{}, kf[(c + 1) & 1]);
        lgkm_release<G>(kf[c & 1]);          // chunk c landed; chunk c+1 stays in flight
      } else {
        lgkm_release<0>(kf[c & 1]);
      }
      static_for<0, G>([&](auto J) {
        constexpr int j = c * G + decltype(J)::value, kb = j / KS, ks = j % KS;
        if constexpr (ks == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
        }
        s[kb] = mfma32(kf[c & 1][decltype(J)::value], qf[ks], s[kb]);
      });
      __builtin_amdgcn_sched_barrier(0);   // keep chunk c's MFMAs here: they cover chunk c+1's flight
    });
    STAMP(8 + t * 8 + 2);
    const int kbase = t * KT;
    if (kbase + KT > a.Nk) {  // ragged last tile: rows >= Nk hold re-read data
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbase + kb * 32 + acc_row(r, lane) >= a.Nk) s[kb][r] = -INFINITY;
    }
    // ---- online softmax: lane = one query, its two halves hold disjoint key rows ---------
    float mx = max3(s[0][0], s[1][0], s[0][1]);
    mx = max3(mx, s[1][1], s[0][2]);
#pragma unroll
    for (int r = 2; r < 15; ++r) mx = max3(mx, s[1][r], s[0][r + 1]);
    mx = max3(mx, s[1][15], __shfl_xor(max3(mx, s[1][15], mx), 32, 64)) * c;
    // defer-max: only re-base when the max grew by more than 2^RESCALE_THR; until then P is
    // bounded by 2^THR instead of 1, which fp32 accumulation absorbs (cdna guide T13).  The
    // previous tile's P.V is complete at this point, so O and l carry exactly one scale.
    constexpr float RESCALE_THR = 6.0f;
    if (!__all(mx - m_run <= RESCALE_THR)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = fast_exp2(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] *= alpha;
    }
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(s[kb][r] * c - m_run);
        s[kb][r] = p;
        rs += p;
      }
    l_run += rs;
    STAMP(8 + t * 8 + 3);
    // ---- O^T += V^T . P^T: group g+1's fragments are read while group g multiplies -------
    __builtin_amdgcn_sched_barrier(0);   // no compiler-issued LDS op may slip between my counted waits
    static_for<0, 4>([&](auto Gi) {
      constexpr int g = decltype(Gi)::value, kb = g >> 1, sp = g & 1;
      if constexpr (g + 1 < 4) {
        issue_v(Int<g + 1>{}, vt[(g + 1) & 1]);
        lgkm_release<6>(vt[g & 1]);
      } else {
        lgkm_release<0>(vt[g & 1]);
      }
      const bf16x8_t pf = acc_to_frag(s[kb], sp);
#pragma unroll
      for (int j = 0; j < 3; ++j)
        o[j] = mfma32(make_bf16x8(vt[g & 1][2 * j], vt[g & 1][2 * j + 1]), pf, o[j]);
      __builtin_amdgcn_sched_barrier(0);
    });
    STAMP(8 + t * 8 + 4);
  }
#ifdef SVIT_ATTN_STAMPS
  if (stamp_on) { g_attn_stamps[2] = __builtin_readcyclecounter(); g_attn_stamps[3] = wall_clock64(); }
#endif

  // ---- epilogue: normalise, add the pooled query (residual pooling), merge heads ----------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  if (qi < a.Nq) {
    if (hh == 0) a.lse2[(size_t)bh * a.Nq + qi] = m_run + log2f(l_tot);
    bf16_t* out = (bf16_t*)a.ctx + ((size_t)b * a.Nq + qi) * a.heads * HD + head * HD;
    const bf16_t* qres = qa + (size_t)qi * DA;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dv = j * 32 + 8 * g + 4 * hh;
        float v0 = o[j][4 * g] * inv, v1 = o[j][4 * g + 1] * inv, v2 = o[j][4 * g + 2] * inv,
              v3 = o[j][4 * g + 3] * inv;
        if (qi > 0) {
          const uint2 qq = *(const uint2*)(qres + dv);
          v0 += lo_bf16(qq.x); v1 += hi_bf16(qq.x); v2 += lo_bf16(qq.y); v3 += hi_bf16(qq.y);
        }
        uint2 pk;
        pk.x = pack_bf16x2(v0, v1);
        pk.y = pack_bf16x2(v2, v3);
        *(uint2*)(out + dv) = pk;
      }
  }
}

template <int DA, int NW, int NS>
int launch_cfg(const svit_attn_fwd_args& a, hipStream_t st) {
  const size_t lds = NS * (size_t)(KT * DA * 2 + KT * HD * 2);
  static SvitOnce once;
  if (int rc = svit_max_lds_once(once, (const void*)attn_fwd_kernel<DA, NW, NS>, lds)) return rc;
  dim3 grid((a.Nq + NW * 32 - 1) / (NW * 32), a.B * a.heads);
  hipLaunchKernelGGL((attn_fwd_kernel<DA, NW, NS>), grid, dim3(NW * 64), lds, st, a);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

template <int DA>
int launch_fwd(const svit_attn_fwd_args& a, hipStream_t st) {
  // 8-wave workgroups once they still cover the chip (one per CU); SVIT_ATTN_FWD_NW forces 4 / 8
  static const int force = getenv("SVIT_ATTN_FWD_NW") ? atoi(getenv("SVIT_ATTN_FWD_NW")) : 0;
  // measured (tools/bench_kernels.py attn): the 8-wave form wins 3-4 % on the long-key blocks
  // (Nk = 1633, DA = 160) and loses on the short ones, where the 8-wave barrier dominates
  const long wg8 = (long)((a.Nq + 255) / 256) * a.B * a.heads;
  const bool wide = force ? force == 8 : (DA == 160 && wg8 >= 200);
  return wide ? launch_cfg<DA, 8, 3>(a, st) : launch_cfg<DA, 4, 2>(a, st);
}
}  // namespace

int svit_attn_fwd_v2(const svit_attn_fwd_args& a, int bias_cols, hipStream_t st);   // attn_fwd2.hip

extern "C" int svit_attn_fwd(const svit_attn_fwd_args* a, void* stream) {
  if (!a || !a->qa || !a->ka || !a->v || !a->ctx || !a->lse2) return SVIT_ERR_ARG;
  if (a->B <= 0 || a->heads <= 0 || a->Nq <= 0 || a->Nk <= 0) return SVIT_ERR_SHAPE;
  if (((uintptr_t)a->qa | (uintptr_t)a->ka | (uintptr_t)a->v | (uintptr_t)a->ctx) & 15)
    return SVIT_ERR_ALIGN;
  if (a->DA != 128 && a->DA != 160) return SVIT_ERR_SHAPE;
  if (a->bias_cols < 0 || a->bias_cols > a->DA - 96) return SVIT_ERR_ARG;
  // SVIT_ATTN_FWD_V=1 selects the round-1 kernel (A/B measurements)
  static const int version = getenv("SVIT_ATTN_FWD_V") ? atoi(getenv("SVIT_ATTN_FWD_V")) : 2;
  if (version != 1) return svit_attn_fwd_v2(*a, a->bias_cols, (hipStream_t)stream);
  if (a->DA == 128) return launch_fwd<128>(*a, (hipStream_t)stream);
  if (a->DA == 160) return launch_fwd<160>(*a, (hipStream_t)stream);
  return SVIT_ERR_SHAPE;
}
