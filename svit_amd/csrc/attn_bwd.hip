// Fused pooled attention, backward (flash-style recompute) -- gfx950.
//
// Two launches (both on the caller's stream):
//   1. dq kernel : first delta[q] = sum_c dctx[q,c] * (ctx[q,c] - q_pooled[q,c]) (rowsum(dO*O)),
//                  kept in a register and stored as (lse2, delta) pairs in the caller's scratch
//                  for launch 2; then per 32-query wave, sweep K/V tiles: S^T, P^T = exp2(c*S - lse2),
//                  dP^T = V dO^T, dS^T = P^T (dP^T - delta) * scale, dQa^T += Ka^T dS^T.
//                  Query on the lane => P/dS reach the next MFMA as B operands in registers.
//   2. dkv kernel: per 32-key wave (128 keys per block), sweep 32-query tiles of a query
//                  chunk: S, P, dP, dS with the KEY on the lane; dV^T += dO^T P, dK^T += Q^T dS
//                  accumulate in registers over the whole sweep; chunks of the query range run
//                  in different blocks and meet in fp32 atomics shaped as whole 384-byte rows
//                  (transposed through LDS first -- row-per-lane atomics are ~17x slower).
// No N x N matrix is ever stored.  The residual-pooling path (ctx += q) contributes dctx to dq
// outside these kernels (svit_pool_ln_bwd's d_res input).
#include <atomic>
#include "attn_common.h"
#include "../../include/svit_hip.h"

namespace {
using namespace attn;
constexpr int KT = 64;   // keys per tile (dq kernel)
constexpr int QT = 32;   // queries per tile (dkv kernel)

// ---------------------------------------------------------------------------------------
// dq kernel.  K/V tiles travel HBM -> LDS by LDS-DMA (no staging registers), two stages, one
// raw barrier per tile behind a counted vmcnt; 2 blocks per CU (2 waves per SIMD) so one
// wave's exp/convert VALU work overlaps the other's MFMAs.
template <int DA>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(svit_attn_bwd_args a, int zero_dkv) {
  constexpr int KS = DA / 16, NP = DA / 32;
  constexpr int K_BYTES = KT * DA * 2, V_BYTES = KT * HD * 2, STAGE = K_BYTES + V_BYTES;
  using KLoad = GldsTile<KT, DA, 4>;
  using VLoad = GldsTile<KT, HD, 4>;
  constexpr int PER_TILE = KLoad::PER_WAVE + VLoad::PER_WAVE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  if (zero_dkv) {   // the dkv launch that follows accumulates with atomics: clear dk / dv here
    const size_t n4 = (size_t)a.B * a.heads * a.Nk * HD / 4;
    const size_t nthr = (size_t)gridDim.x * gridDim.y * 256;
    const size_t me = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid;
    for (size_t i = me; i < n4; i += nthr) {
      ((float4*)a.dk)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      ((float4*)a.dv)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const int wgid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int bh = wgid / gridDim.x, b = bh / a.heads, head = bh % a.heads;
  const int qtile = wgid % gridDim.x;
  const int qi = qtile * 128 + wave * 32 + (lane & 31);
  const int qc = min(qi, a.Nq - 1);
  const bf16_t* qa = (const bf16_t*)a.qa + ((size_t)bh * a.Nq) * DA;
  const bf16_t* ka = (const bf16_t*)a.ka + ((size_t)bh * a.Nk) * DA;
  const bf16_t* vv = (const bf16_t*)a.v + ((size_t)bh * a.Nk) * HD;
  const bf16_t* dor = (const bf16_t*)a.dctx + ((size_t)b * a.Nq + qc) * a.heads * HD + head * HD;
  const float c = a.scale * 1.4426950408889634f;
  const float lse = a.lse2[(size_t)bh * a.Nq + qc];

  bf16x8_t qf[KS], dof[6];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8_t*)(qa + (size_t)qc * DA + ks * 16 + hh * 8);
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) dof[ks] = *(const bf16x8_t*)(dor + ks * 16 + hh * 8);
  // delta = rowsum(dO . O) with O = ctx - q (residual pooling adds q to every token but cls):
  // folded in here (was a separate pre-pass); each half-wave lane holds 48 of the 96 channels.
  // The (lse2, delta) pair is also what the dkv kernel streams, so it is written back once.
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
    if (hh == 0 && qi < a.Nq) ((float2*)a.delta)[(size_t)bh * a.Nq + qi] = make_float2(lse, dlt);
  }

  // Pin the register operands NOW: their first use must not sit inside the tile loop, or the
  // compiler's wait for them (vmcnt(0)) would drain the LDS-DMA pipeline every iteration.
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) asm volatile("" : "+v"(dof[ks]));
  float lse_p = lse, dlt_p = dlt;
  asm volatile("" : "+v"(lse_p), "+v"(dlt_p));

  f32x16_t dq[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[j][r] = 0.f;

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
    const int kbase = t * KT;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16_t s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s = mfma32(row_frag<KT>(k_cur, kb * 32, ks, lane), qf[ks], s);
#pragma unroll
      for (int ks = 0; ks < 6; ++ks) dp = mfma32(row_frag<KT>(v_cur, kb * 32, ks, lane), dof[ks], dp);
      if (kbase + KT > a.Nk) {   // ragged last tile: rows past Nk hold re-read data
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbase + kb * 32 + acc_row(r, lane) >= a.Nk) s[r] = -INFINITY;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r] * c - lse_p) * (dp[r] - dlt_p) * a.scale;
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        const bf16x8_t dsf = acc_to_frag(s, sp);
        bf16x8_t kt[NP];
        tr_frags_asm<KT, NP>(k_cur, kb * 32 + sp * 16, lane, kt);
#pragma unroll
        for (int j = 0; j < NP; ++j) dq[j] = mfma32(kt[j], dsf, dq[j]);
      }
    }
  }
  if (qi < a.Nq) {
    bf16_t* out = (bf16_t*)a.dqa + ((size_t)bh * a.Nq + qi) * DA;
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 pk;
        pk.x = pack_bf16x2(dq[j][4 * g], dq[j][4 * g + 1]);
        pk.y = pack_bf16x2(dq[j][4 * g + 2], dq[j][4 * g + 3]);
        *(uint2*)(out + j * 32 + 8 * g + 4 * hh) = pk;
      }
  }
}

// ---------------------------------------------------------------------------------------
// dkv kernel.  Q / dO tiles and the (lse2, delta) pairs arrive by LDS-DMA into a three-stage
// ring; key on the lane; dK^T / dV^T accumulate in registers over the sweep.
// NH = 1: 4 waves, 32-query tiles.  NH = 2 (round 2): 8 waves -- the four key groups twice, the
// two halves working on the two 32-query halves of a 64-query tile -- so that a CU that holds
// ONE workgroup (the split heuristic aims at one per CU: every extra split costs a full fp32
// atomic flush) still runs two waves per SIMD: one half's exp / convert / LDS waits hide behind
// the other's MFMAs.  The halves' dK / dV meet in LDS before the (unchanged) row stores.
template <int DA, int NH>
__global__ __launch_bounds__(256 * NH, NH == 2 ? 1 : 2) void attn_bwd_dkv_kernel(svit_attn_bwd_args a,
                                                                                 int tiles_per_split) {
  constexpr int KS = DA / 16;
  constexpr int QR = QT * NH;                             // query rows per ring stage
  constexpr int Q_BYTES = QR * DA * 2, O_BYTES = QR * HD * 2;
  constexpr int STAGE = Q_BYTES + O_BYTES + 2 * QR * 4;   // [Q | dO | (lse2, delta) pairs]
  constexpr int NSTAGE = 3;
  constexpr int OUT_LD = HD + 1;                          // padded fp32 transpose buffer
  constexpr int NTHR = 256 * NH;
  using QLoad = GldsTile<QR, DA, 4 * NH>;
  using OLoad = GldsTile<QR, HD, 4 * NH>;
  constexpr int PER_TILE = QLoad::PER_WAVE + OLoad::PER_WAVE + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int kg = wave & 3, qh = wave >> 2;                // key group, query half
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
  const float* ld_g = a.delta + (size_t)bh * a.Nq * 2;    // (lse2, delta) pairs
  const float c = a.scale * 1.4426950408889634f;

  bf16x8_t kf[KS], vf[6];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kf[ks] = *(const bf16x8_t*)(ka + (size_t)kc * DA + ks * 16 + hh * 8);
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) vf[ks] = *(const bf16x8_t*)(vv + (size_t)kc * HD + ks * 16 + hh * 8);

  // pin the register operands before the tile loop (see the dq kernel)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(kf[ks]));
#pragma unroll
  for (int ks = 0; ks < 6; ++ks) asm volatile("" : "+v"(vf[ks]));

  f32x16_t dk[3], dv[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[j][r] = 0.f; dv[j][r] = 0.f; }

  const int nqt = (a.Nq + QR - 1) / QR;
  const int t_begin = by * tiles_per_split;
  const int t_end = min(nqt, t_begin + tiles_per_split);
  QLoad qload;
  OLoad oload;
  qload.init(DA, wave, lane);
  oload.init((size_t)a.heads * HD, wave, lane);
  auto issue = [&](int t) {
    unsigned char* st = smem + ((t - t_begin) % NSTAGE) * STAGE;
    const int q0 = t * QR;
    qload.issue_auto(qa + (size_t)q0 * DA, DA, a.Nq - q0, st, wave, lane);
    oload.issue_auto(dob + (size_t)q0 * a.heads * HD, (size_t)a.heads * HD, a.Nq - q0, st + Q_BYTES, wave, lane);
    // 32 (lse2, delta) pairs = 64 floats = one dword LDS-DMA (rows past Nq re-read the last pair);
    // with two halves the even waves fetch the first 32 pairs, the odd waves the second 32
    const int part = NH == 2 ? (wave & 1) : 0;
    const int qrow = min(q0 + part * 32 + (lane >> 1), a.Nq - 1);
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(ld_g + (size_t)qrow * 2 + (lane & 1)),
        (__attribute__((address_space(3))) void*)(st + Q_BYTES + O_BYTES + part * 256), 4, 0, 0);
  };
  if (t_begin < t_end) issue(t_begin);
  if (t_begin + 1 < t_end) issue(t_begin + 1);
  for (int t = t_begin; t < t_end; ++t) {
    if (t + 1 < t_end) wait_vmcnt<PER_TILE>();   // all but the youngest tile's loads are done
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + 2 < t_end) issue(t + 2);             // into the stage consumed two steps ago
    const unsigned char* q_cur = smem + ((t - t_begin) % NSTAGE) * STAGE;
    const unsigned char* o_cur = q_cur + Q_BYTES;
    const float* ld_s = (const float*)(o_cur + O_BYTES) + qh * 64;
    const int q0 = t * QR + qh * QT;             // first query of this half's 32 rows
    const int r0 = qh * QT;                      // their row offset inside the staged tile

    f32x16_t s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) s = mfma32(row_frag<QR>(q_cur, r0, ks, lane), kf[ks], s);
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) dp = mfma32(row_frag<QR>(o_cur, r0, ks, lane), vf[ks], dp);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      // rows 8g + 4hh + e, e = 0..3: four consecutive (lse2, delta) pairs
      const float4 p01 = *(const float4*)(ld_s + 2 * (8 * g + 4 * hh));
      const float4 p23 = *(const float4*)(ld_s + 2 * (8 * g + 4 * hh) + 4);
      const float lv[4] = {p01.x, p01.z, p23.x, p23.z}, dl[4] = {p01.y, p01.w, p23.y, p23.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float p = fast_exp2(s[4 * g + e] * c - lv[e]);
        if (q0 + QT > a.Nq && q0 + 8 * g + 4 * hh + e >= a.Nq) p = 0.f;   // re-read rows past Nq
        s[4 * g + e] = p;
        dp[4 * g + e] = p * (dp[4 * g + e] - dl[e]) * a.scale;
      }
    }
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      const bf16x8_t pf = acc_to_frag(s, sp);
      const bf16x8_t dsf = acc_to_frag(dp, sp);
      bf16x8_t ot[3], qt[3];
      tr_frags_asm<QR, 3>(o_cur, r0 + sp * 16, lane, ot);
#pragma unroll
      for (int j = 0; j < 3; ++j) dv[j] = mfma32(ot[j], pf, dv[j]);
      tr_frags_asm<QR, 3>(q_cur, r0 + sp * 16, lane, qt);
#pragma unroll
      for (int j = 0; j < 3; ++j) dk[j] = mfma32(qt[j], dsf, dk[j]);
    }
  }
  // ---- transpose through LDS so that every atomic wave-instruction adds whole rows; with two
  // ---- query halves the second half adds its accumulators to the first half's in the buffer
  float* obuf = (float*)smem;  // [128 keys][OUT_LD]
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    if (qh == 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          obuf[(kg * 32 + (lane & 31)) * OUT_LD + j * 32 + acc_row(r, lane)] =
              pass == 0 ? dk[j][r] : dv[j][r];
    }
    if (NH == 2) {
      __syncthreads();
      if (qh == 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            obuf[(kg * 32 + (lane & 31)) * OUT_LD + j * 32 + acc_row(r, lane)] +=
                pass == 0 ? dk[j][r] : dv[j][r];
      }
    }
    __syncthreads();
    float* dst = (pass == 0 ? a.dk : a.dv) + ((size_t)bh * a.Nk) * HD;
    if (gridDim.y == 1) {   // this block owns its keys outright: plain coalesced row stores
      for (int i = tid; i < 128 * HD; i += NTHR) {
        const int kr = i / HD, d = i % HD;
        if (key0 + kr < a.Nk) dst[(size_t)(key0 + kr) * HD + d] = obuf[kr * OUT_LD + d];
      }
    } else {
      for (int i = tid; i < 128 * HD; i += NTHR) {
        const int kr = i / HD, d = i % HD;
        if (key0 + kr < a.Nk) atomicAdd(dst + (size_t)(key0 + kr) * HD + d, obuf[kr * OUT_LD + d]);
      }
    }
  }
}

static std::atomic<int> g_dkv_halves{0};   // tuning knob (svit_attn_debug_set(0, n)): 1 / 2, 0 = heuristic

template <int DA>
int launch_bwd(const svit_attn_bwd_args& a, hipStream_t st) {
  static SvitOnce once_dq, once_kv, once_kv2;
  const size_t lds_dq = 2 * (size_t)(KT * DA * 2 + KT * HD * 2);
  const size_t lds_out = (size_t)128 * (HD + 1) * 4;
  size_t lds_kv = 3 * (size_t)(QT * DA * 2 + QT * HD * 2 + 2 * QT * 4);
  size_t lds_kv2 = 3 * (size_t)(2 * QT * DA * 2 + 2 * QT * HD * 2 + 4 * QT * 4);
  if (lds_kv < lds_out) lds_kv = lds_out;
  if (lds_kv2 < lds_out) lds_kv2 = lds_out;
  if (int rc = svit_max_lds_once(once_dq, (const void*)attn_bwd_dq_kernel<DA>, lds_dq)) return rc;
  if (int rc = svit_max_lds_once(once_kv, (const void*)attn_bwd_dkv_kernel<DA, 1>, lds_kv)) return rc;
  if (int rc = svit_max_lds_once(once_kv2, (const void*)attn_bwd_dkv_kernel<DA, 2>, lds_kv2)) return rc;
  const int key_blocks = (a.Nk + 127) / 128;
  const int base = key_blocks * a.B * a.heads;
  // two query halves (8 waves) where the launch leaves one 4-wave workgroup per CU anyway: the
  // short-key blocks (measured, tools/bench_kernels.py attn, profiles/r02_attn_bwd_dkv_halves.txt:
  // -8..-16 % of the whole backward at Nk = 457; +6..13 % at Nk = 1633 / DA = 160, whose 8-wave
  // form sits at the 256-VGPR limit with a 100 KB ring)
  int halves = g_dkv_halves.load();
  if (halves != 1 && halves != 2) halves = (DA == 128 && base <= 256) ? 2 : 1;
  const int nqt = (a.Nq + QT * halves - 1) / (QT * halves);
  int splits = a.q_splits;
  if (splits <= 0) {
    // every split adds a full [128 keys x 192] fp32 tile per block with atomics (~1.3 TB/s
    // chip-wide), so split the query range only as far as needed to fill the chip
    // (measured, tools/bench_kernels.py attnsplits: ~1 block per CU is the sweet spot), and
    // keep >= 4 query tiles per block to amortise the epilogue
    splits = (256 + base - 1) / base;
    if (splits > nqt / 4) splits = nqt / 4;
  }
  if (splits > nqt) splits = nqt;
  if (splits < 1) splits = 1;
  int tiles_per_split = (nqt + splits - 1) / splits;
  splits = (nqt + tiles_per_split - 1) / tiles_per_split;
  if (((uintptr_t)a.dk | (uintptr_t)a.dv) & 15) return SVIT_ERR_ALIGN;
  hipLaunchKernelGGL(attn_bwd_dq_kernel<DA>, dim3((a.Nq + 127) / 128, a.B * a.heads), dim3(256),
                     lds_dq, st, a, splits > 1 ? 1 : 0);
  SVIT_LAUNCH_CHECK();
  if (halves == 2)
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DA, 2>), dim3(key_blocks, splits, a.B * a.heads), dim3(512),
                       lds_kv2, st, a, tiles_per_split);
  else
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DA, 1>), dim3(key_blocks, splits, a.B * a.heads), dim3(256),
                       lds_kv, st, a, tiles_per_split);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}
}  // namespace

extern "C" int svit_attn_debug_set(int key, int val) {
  if (key == 0) g_dkv_halves = val;
  else return SVIT_ERR_ARG;
  return SVIT_OK;
}

extern "C" int svit_attn_bwd(const svit_attn_bwd_args* a, void* stream) {
  if (!a || !a->qa || !a->ka || !a->v || !a->ctx || !a->dctx || !a->lse2 || !a->delta || !a->dqa ||
      !a->dk || !a->dv)
    return SVIT_ERR_ARG;
  if (a->B <= 0 || a->heads <= 0 || a->Nq <= 0 || a->Nk <= 0) return SVIT_ERR_SHAPE;
  if (a->B * a->heads > 65535) return SVIT_ERR_SHAPE;
  if (a->DA == 128) return launch_bwd<128>(*a, (hipStream_t)stream);
  if (a->DA == 160) return launch_bwd<160>(*a, (hipStream_t)stream);
  return SVIT_ERR_SHAPE;
}
