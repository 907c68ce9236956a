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
#include "attn_common.h"
#include "../../include/svit_hip.h"

namespace {
using namespace attn;
constexpr int KT = 64;  // keys per tile

template <int DA>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(svit_attn_fwd_args a) {
  constexpr int KS = DA / 16;                 // k-steps of the QK^T contraction
  constexpr int K_BYTES = KT * DA * 2, V_BYTES = KT * HD * 2;
  constexpr int STAGE = K_BYTES + V_BYTES;    // [K tile | V tile] per pipeline stage
  using KLoad = GldsTile<KT, DA, 4>;
  using VLoad = GldsTile<KT, HD, 4>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int wgid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int bh = wgid / gridDim.x, b = bh / a.heads, head = bh % a.heads;
  const int q0 = (wgid % gridDim.x) * 128 + wave * 32;
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

  const int nt = (a.Nk + KT - 1) / KT;
  KLoad kload;
  VLoad vload;
  kload.init(DA, wave, lane);
  vload.init(HD, wave, lane);
  auto issue = [&](int t) {
    unsigned char* st = smem + (t & 1) * STAGE;
    const int k0 = t * KT;
    kload.issue_auto(ka + (size_t)k0 * DA, DA, a.Nk - k0, st, wave, lane);
    vload.issue_auto(vv + (size_t)k0 * HD, HD, a.Nk - k0, st + K_BYTES, wave, lane);
  };
  issue(0);
  for (int t = 0; t < nt; ++t) {
    wait_vmcnt<0>();                 // this wave's share of tile t has landed
    __builtin_amdgcn_s_barrier();    // everyone's share has; everyone is done with tile t-1
    if (t + 1 < nt) issue(t + 1);    // travels while tile t is consumed
    const unsigned char* k_cur = smem + (t & 1) * STAGE;
    const unsigned char* v_cur = k_cur + K_BYTES;
    // ---- S^T = ka . qa^T for the two 32-key blocks of the tile --------------------------
    f32x16_t s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        s[kb] = mfma32(row_frag<KT>(k_cur, kb * 32, ks, lane), qf[ks], s[kb]);
    }
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
    // ---- O^T += V^T . P^T -------------------------------------------------------------
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        const bf16x8_t pf = acc_to_frag(s[kb], sp);
        bf16x8_t vt[3];
        tr_frags_asm<KT, 3>(v_cur, kb * 32 + sp * 16, lane, vt);
#pragma unroll
        for (int j = 0; j < 3; ++j) o[j] = mfma32(vt[j], pf, o[j]);
      }
  }

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

template <int DA>
int launch_fwd(const svit_attn_fwd_args& a, hipStream_t st) {
  const size_t lds = 2 * (size_t)(KT * DA * 2 + KT * HD * 2);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel<DA>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  dim3 grid((a.Nq + 127) / 128, a.B * a.heads);
  hipLaunchKernelGGL(attn_fwd_kernel<DA>, grid, dim3(256), lds, st, a);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}
}  // namespace

extern "C" int svit_attn_fwd(const svit_attn_fwd_args* a, void* stream) {
  if (!a || !a->qa || !a->ka || !a->v || !a->ctx || !a->lse2) return SVIT_ERR_ARG;
  if (a->B <= 0 || a->heads <= 0 || a->Nq <= 0 || a->Nk <= 0) return SVIT_ERR_SHAPE;
  if (((uintptr_t)a->qa | (uintptr_t)a->ka | (uintptr_t)a->v | (uintptr_t)a->ctx) & 15)
    return SVIT_ERR_ALIGN;
  if (a->DA == 128) return launch_fwd<128>(*a, (hipStream_t)stream);
  if (a->DA == 160) return launch_fwd<160>(*a, (hipStream_t)stream);
  return SVIT_ERR_SHAPE;
}
