// Shared epilogue of the NT GEMM kernels (gemm.hip, gemm_v2.hip) -- gfx950.
#pragma once
#include "common.h"
#ifndef SVIT_EPI_AHEAD      // slabs of epilogue operand lookahead: 1; 2 (a ring of three register sets) measured level in the step (round 4)
#define SVIT_EPI_AHEAD 1
#endif
#include "../../include/svit_hip.h"

// Each wave transposes its accumulators, 16 rows x (32*NB) columns at a time, through a
// private LDS region so that global traffic is row-contiguous and vectorised (16 B fp32 / 8 B
// bf16 per lane) instead of 2-byte column-strided accesses.  Needs
// n_waves * 16 * (32*NB + 4) * 4 bytes of LDS at `smem`; all waves of the block must call it.
// bf16-output epilogues (plain, GELU, GELU-backward) with 8 columns = one 16-byte store per lane
// (the 4-column form below issues twice as many 8-byte stores; the stage-1 GEMMs write 150-310 MB
// per launch and are bound by exactly that).  Needs 16-byte aligned rows: ldo / ldo2 / ldaux % 8.
// WAVE_PRIVATE: the staging region belongs to the calling wave alone and the LDS operations of one wave
// complete in order, so a drained lgkmcnt (plus the compiler barrier of the asm) is all the
// synchronisation the write -> read -> overwrite sequence needs; callers whose OTHER waves have left the
// kernel or are elsewhere (gemm_nt_ring_kernel's loader waves) must use it -- no s_barrier is executed.
template <bool WAVE_PRIVATE>
__device__ __forceinline__ void nt_epi_sync() {
  if constexpr (WAVE_PRIVATE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  else __syncthreads();
}

template <int RB, int NB, int EPI, bool WAVE_PRIVATE = false>
__device__ __forceinline__ void nt_epilogue_wide(const svit_gemm_args& p, f32x16_t (&acc)[RB][NB],
                                                 unsigned char* smem, int m0, int n0, int wm, int wn,
                                                 int lane, int wave) {
  constexpr int WN = 32 * NB, EP_LD = WN + 4, GPR = 4 * NB;     // 8-column groups per row
  float* stg = (float*)smem + wave * (16 * EP_LD);
  // the saved gelu'(h) slabs are fetched SVIT_EPI_AHEAD 16-row slabs ahead through a ring of three register sets.  Two
  // ahead was tried in round 4 (inside the step the slabs come from HBM, not from the Infinity Cache of an isolated
  // loop): 12.66 / 12.68 vs 12.66 / 12.72 ms per step -- level; one stays (fewer live registers)
  uint4 aux_ring[3][NB];
  auto fetch_aux = [&](int ih, uint4 (&dst)[NB]) {
    const int i = ih >> 1, half = ih & 1;
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int idx = lane + 64 * it, rl = idx / GPR, c8 = idx % GPR;
      const int row = m0 + wm * 32 * RB + i * 32 + half * 16 + rl, col = n0 + wn * WN + c8 * 8;
      dst[it] = make_uint4(0, 0, 0, 0);
      if (row < p.M && col < p.N) dst[it] = *(const uint4*)((const bf16_t*)p.aux + (size_t)row * p.ldaux + col);
    }
  };
  float4 bias_r[NB][2];
#pragma unroll
  for (int it = 0; it < NB; ++it) {
    const int col = n0 + wn * WN + ((lane + 64 * it) % GPR) * 8;
    bias_r[it][0] = bias_r[it][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && col < p.N) {
      bias_r[it][0] = *(const float4*)(p.bias + col);
      bias_r[it][1] = *(const float4*)(p.bias + col + 4);
    }
  }
  constexpr int AHEAD = SVIT_EPI_AHEAD;
  if (EPI == SVIT_EPI_DGELU) {
    fetch_aux(0, aux_ring[0]);
    if (AHEAD > 1 && 2 * RB > 1) fetch_aux(1, aux_ring[1]);
  }
#pragma unroll
  for (int ih = 0; ih < 2 * RB; ++ih) {
    const int i = ih >> 1, half = ih & 1;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int rr = 0; rr < 8; ++rr)
        stg[((rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5)) * EP_LD + j * 32 + (lane & 31)] =
            acc[i][j][half * 8 + rr];
    if (EPI == SVIT_EPI_DGELU && ih + AHEAD < 2 * RB) fetch_aux(ih + AHEAD, aux_ring[(ih + AHEAD) % 3]);
    nt_epi_sync<WAVE_PRIVATE>();
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int idx = lane + 64 * it, rl = idx / GPR, c8 = idx % GPR;
      const int row = m0 + wm * 32 * RB + i * 32 + half * 16 + rl, col = n0 + wn * WN + c8 * 8;
      if (row >= p.M || col >= p.N) continue;
      const float4 v0 = *(const float4*)(stg + rl * EP_LD + c8 * 8), v1 = *(const float4*)(stg + rl * EP_LD + c8 * 8 + 4);
      const float4 b0 = bias_r[it][0], b1 = bias_r[it][1];
      float v[8] = {v0.x + b0.x, v0.y + b0.y, v0.z + b0.z, v0.w + b0.w, v1.x + b1.x, v1.y + b1.y, v1.z + b1.z, v1.w + b1.w};
      uint4 o;
      if constexpr (EPI == SVIT_EPI_BF16) {
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
      } else if constexpr (EPI == SVIT_EPI_GELU) {
        float a[8], d[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gelu_fwd_grad(v[e], &a[e], &d[e]);
        if (p.out2) {
          uint4 o2;
          o2.x = pack_bf16x2(d[0], d[1]); o2.y = pack_bf16x2(d[2], d[3]);
          o2.z = pack_bf16x2(d[4], d[5]); o2.w = pack_bf16x2(d[6], d[7]);
          *(uint4*)((bf16_t*)p.out2 + (size_t)row * p.ldo2 + col) = o2;
        }
        o.x = pack_bf16x2(a[0], a[1]); o.y = pack_bf16x2(a[2], a[3]);
        o.z = pack_bf16x2(a[4], a[5]); o.w = pack_bf16x2(a[6], a[7]);
      } else {   // SVIT_EPI_DGELU: acc * saved gelu'(h)
        const uint4 h = aux_ring[ih % 3][it];
        o.x = pack_bf16x2(v[0] * lo_bf16(h.x), v[1] * hi_bf16(h.x));
        o.y = pack_bf16x2(v[2] * lo_bf16(h.y), v[3] * hi_bf16(h.y));
        o.z = pack_bf16x2(v[4] * lo_bf16(h.z), v[5] * hi_bf16(h.z));
        o.w = pack_bf16x2(v[6] * lo_bf16(h.w), v[7] * hi_bf16(h.w));
      }
      *(uint4*)((bf16_t*)p.out + (size_t)row * p.ldo + col) = o;
    }
    if (ih + 1 < 2 * RB) nt_epi_sync<WAVE_PRIVATE>();
  }
}

// SVIT_EPI_RELQ: the product q . Rcat^T is consumed where it is produced -- every (row, j) of the query
// side's rel-pos columns picks its table row out of the staged 16-row slab (relq_map: the column of the
// product for (token, j), -1 = zero) and goes to qa[row, 96 + j].  The product itself is never stored.
// A column tile writes the entries whose column it owns; the tile at column 0 also writes the zeros.
template <int RB, int NB>
__device__ __forceinline__ void nt_epilogue_relq(const svit_gemm_args& p, f32x16_t (&acc)[RB][NB],
                                                 unsigned char* smem, int m0, int n0, int wm, int wn,
                                                 int lane, int wave) {
  constexpr int WN = 32 * NB, EP_LD = WN + 4;
  constexpr int MAXV = 16;                                          // (row, j) entries per lane and slab: 16 * 64 / 64
  float* stg = (float*)smem + wave * (16 * EP_LD);
  const int extra = p.relq_extra, sh = extra == 64 ? 6 : 5;        // 32 or 64 columns per row
  const int nv = 16 * extra / 64;                                   // 8 or 16
  const int c_lo = n0 + wn * WN;
  bf16_t* qa = (bf16_t*)p.relq_out;
  // every map entry this wave will need, requested before anything is staged: one memory round trip for
  // all slabs instead of one per slab in front of its stores
  int cols[2 * RB][MAXV];
#pragma unroll
  for (int ih = 0; ih < 2 * RB; ++ih) {
    const int row0 = m0 + wm * 32 * RB + (ih >> 1) * 32 + (ih & 1) * 16;
    const int tok0 = row0 % p.relq_rows;
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
      const int e = it * 64 + lane, rl = e >> sh, j = e & (extra - 1);
      cols[ih][it] = -2;                                            // -2: no such entry / row past M
      if (it < nv && row0 + rl < p.M) {
        int tok = tok0 + rl;
        while (tok >= p.relq_rows) tok -= p.relq_rows;              // (once at most unless a (b, head) has < 16 tokens)
        cols[ih][it] = p.relq_map[tok * extra + j];
      }
    }
  }
#pragma unroll
  for (int ih = 0; ih < 2 * RB; ++ih) {
    const int i = ih >> 1, half = ih & 1;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int rr = 0; rr < 8; ++rr)
        stg[((rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5)) * EP_LD + j * 32 + (lane & 31)] =
            acc[i][j][half * 8 + rr];
    nt_epi_sync<true>();      // (the staging region is the wave's own)
    const int row0 = m0 + wm * 32 * RB + i * 32 + half * 16;
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
      const int e = it * 64 + lane, rl = e >> sh, j = e & (extra - 1), col = cols[ih][it];
      if (col == -2) continue;
      const size_t o = (size_t)(row0 + rl) * p.relq_ld + 96 + j;
      if (col < 0) {
        if (c_lo == 0) qa[o] = 0;
      } else if (col >= c_lo && col < c_lo + WN) {
        // two roundings, as the unfused pair had (the product stored as bf16, then scaled)
        qa[o] = f32_to_bf16(bf16_to_f32(f32_to_bf16(stg[rl * EP_LD + col - c_lo])) * p.relq_scale);
      }
    }
    if (ih + 1 < 2 * RB) nt_epi_sync<true>();
  }
}

template <int RB, int NB, int EPI, bool WAVE_PRIVATE = false>
__device__ __forceinline__ void nt_epilogue(const svit_gemm_args& p, f32x16_t (&acc)[RB][NB],
                                            unsigned char* smem, int m0, int n0, int wm, int wn,
                                            int lane, int wave) {
  if constexpr (EPI == SVIT_EPI_RELQ) {
    nt_epilogue_relq<RB, NB>(p, acc, smem, m0, n0, wm, wn, lane, wave);
    return;
  }
  if constexpr (EPI == SVIT_EPI_BF16 || EPI == SVIT_EPI_GELU || EPI == SVIT_EPI_DGELU) {
    const bool rows16 = (p.ldo % 8 == 0) && (EPI != SVIT_EPI_GELU || !p.out2 || p.ldo2 % 8 == 0) &&
                        (EPI != SVIT_EPI_DGELU || p.ldaux % 8 == 0);
    if (rows16) {     // (uniform over the launch)
      nt_epilogue_wide<RB, NB, EPI, WAVE_PRIVATE>(p, acc, smem, m0, n0, wm, wn, lane, wave);
      return;
    }
  }
  constexpr int WN = 32 * NB;
  constexpr int EP_LD = WN + 4;
  constexpr int NIT = 2 * NB;
  // epilogues that READ a second operand (residual, saved GELU', fp32 accumulate): its 16-row
  // slab is fetched into registers one slab ahead, so the loads fly while the previous slab is
  // staged / stored instead of exposing one memory round trip per 4 columns
  constexpr bool HAS_AUX = (EPI == SVIT_EPI_RESID || EPI == SVIT_EPI_DGELU || EPI == SVIT_EPI_F32);
  float* stg = (float*)smem + wave * (16 * EP_LD);
  float4 aux_ring[3][NIT];      // two slabs ahead, as in nt_epilogue_wide
  constexpr int AHEAD = SVIT_EPI_AHEAD;
  const bool use_aux = HAS_AUX && (EPI != SVIT_EPI_F32 || p.accumulate);
  auto out_row = [&](int row) -> size_t {
    if (EPI == SVIT_EPI_F32 && p.remap_L > 0)
      return (size_t)(row / p.remap_L) * p.remap_N + p.remap_off + (row % p.remap_L);
    return (size_t)row;
  };
  auto fetch_aux = [&](int ih, float4 (&dst)[NIT]) {
    const int i = ih >> 1, half = ih & 1;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = lane + 64 * it;
      const int rl = idx / (8 * NB), c4 = idx % (8 * NB);
      const int row = m0 + wm * 32 * RB + i * 32 + half * 16 + rl;
      const int col = n0 + wn * WN + c4 * 4;
      dst[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row >= p.M || col >= p.N) continue;
      if constexpr (EPI == SVIT_EPI_RESID) {
        dst[it] = *(const float4*)((const float*)p.aux + (size_t)row * p.ldaux + col);
      } else if constexpr (EPI == SVIT_EPI_DGELU) {
        const uint2 h = *(const uint2*)((const bf16_t*)p.aux + (size_t)row * p.ldaux + col);
        dst[it].x = __uint_as_float(h.x);
        dst[it].y = __uint_as_float(h.y);
      } else if constexpr (EPI == SVIT_EPI_F32) {
        dst[it] = *(const float4*)((const float*)p.out + out_row(row) * p.ldo + col);
      }
    }
  };
  // the bias of this lane's NIT column groups is the same for every 16-row slab: load it once
  float4 bias_r[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int col = n0 + wn * WN + ((lane + 64 * it) % (8 * NB)) * 4;
    bias_r[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && col < p.N) bias_r[it] = *(const float4*)(p.bias + col);
  }
  // DropPath row scale (EPI_RESID): a wave's 32*RB rows span at most two samples when a sample
  // has at least that many rows -- two loads up front instead of one per row in the store loop
  float rs_lo = 1.f, rs_hi = 1.f;
  int rs_boundary = 0x7fffffff;
  const bool rs_fast = EPI == SVIT_EPI_RESID && p.row_scale && p.rows_per_sample >= 32 * RB;
  if (rs_fast) {
    const int r_first = min(m0 + wm * 32 * RB, p.M - 1), r_last = min(r_first + 32 * RB - 1, p.M - 1);
    rs_lo = p.row_scale[r_first / p.rows_per_sample];
    rs_hi = p.row_scale[r_last / p.rows_per_sample];
    rs_boundary = (r_first / p.rows_per_sample + 1) * p.rows_per_sample;
  }
  if (use_aux) {
    fetch_aux(0, aux_ring[0]);
    if (AHEAD > 1 && 2 * RB > 1) fetch_aux(1, aux_ring[1]);
  }
#pragma unroll
  for (int ih = 0; ih < 2 * RB; ++ih) {
    const int i = ih >> 1, half = ih & 1;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int rr = 0; rr < 8; ++rr)
        stg[((rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5)) * EP_LD + j * 32 + (lane & 31)] =
            acc[i][j][half * 8 + rr];
    if (use_aux && ih + AHEAD < 2 * RB) fetch_aux(ih + AHEAD, aux_ring[(ih + AHEAD) % 3]);
    nt_epi_sync<WAVE_PRIVATE>();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = lane + 64 * it;
      const int rl = idx / (8 * NB), c4 = idx % (8 * NB);
      const int row = m0 + wm * 32 * RB + i * 32 + half * 16 + rl;
      const int col = n0 + wn * WN + c4 * 4;
      if (row >= p.M || col >= p.N) continue;
      float4 v = *(const float4*)(stg + rl * EP_LD + c4 * 4);
      {
        const float4 b = bias_r[it];
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      if constexpr (EPI == SVIT_EPI_BF16) {
        uint2 o;
        o.x = pack_bf16x2(v.x, v.y); o.y = pack_bf16x2(v.z, v.w);
        *(uint2*)((bf16_t*)p.out + (size_t)row * p.ldo + col) = o;
      } else if constexpr (EPI == SVIT_EPI_GELU) {
        // the exp / cdf of the forward give the derivative for two more FMAs: it is saved
        // (bf16) instead of the pre-activation, and fc2's dgrad epilogue is one multiply.
        // out2 == NULL (no-grad passes): only gelu(h) is produced.
        float a0, a1, a2, a3, d0, d1, d2, d3;
        gelu_fwd_grad(v.x, &a0, &d0); gelu_fwd_grad(v.y, &a1, &d1);
        gelu_fwd_grad(v.z, &a2, &d2); gelu_fwd_grad(v.w, &a3, &d3);
        uint2 o;
        if (p.out2) {
          o.x = pack_bf16x2(d0, d1); o.y = pack_bf16x2(d2, d3);
          *(uint2*)((bf16_t*)p.out2 + (size_t)row * p.ldo2 + col) = o;
        }
        o.x = pack_bf16x2(a0, a1); o.y = pack_bf16x2(a2, a3);
        *(uint2*)((bf16_t*)p.out + (size_t)row * p.ldo + col) = o;
      } else if constexpr (EPI == SVIT_EPI_RESID) {
        const float s = rs_fast ? (row < rs_boundary ? rs_lo : rs_hi)
                                : (p.row_scale ? p.row_scale[row / p.rows_per_sample] : 1.f);
        const float4 res = aux_ring[ih % 3][it];
        float4 o;
        o.x = res.x + s * v.x; o.y = res.y + s * v.y; o.z = res.z + s * v.z; o.w = res.w + s * v.w;
        *(float4*)((float*)p.out + (size_t)row * p.ldo + col) = o;
      } else if constexpr (EPI == SVIT_EPI_F32) {
        float4* o = (float4*)((float*)p.out + out_row(row) * p.ldo + col);
        if (p.accumulate) {
          const float4 old = aux_ring[ih % 3][it];
          v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
        }
        *o = v;
      } else if constexpr (EPI == SVIT_EPI_DGELU) {
        const uint32_t hx = __float_as_uint(aux_ring[ih % 3][it].x), hy = __float_as_uint(aux_ring[ih % 3][it].y);
        uint2 o;
        o.x = pack_bf16x2(v.x * lo_bf16(hx), v.y * hi_bf16(hx));
        o.y = pack_bf16x2(v.z * lo_bf16(hy), v.w * hi_bf16(hy));
        *(uint2*)((bf16_t*)p.out + (size_t)row * p.ldo + col) = o;
      }
    }
    if (ih + 1 < 2 * RB) nt_epi_sync<WAVE_PRIVATE>();
  }
}
