// Shared pieces of the fused pooled-attention kernels (forward + backward) -- gfx950.
//
// LDS tile image ("panel image"): a [rows][cols] bf16 tile is stored as cols/32 panels; panel p
// holds columns 32p..32p+31 of every row as 64-byte rows, with the four 16-byte chunks of a row
// XOR-swizzled by ((row>>2)&3):
//     byte(row, col) = p*rows*64 + row*64 + 16*((c ^ ((row>>2)&3))) + 2*(col&7),
//     p = col>>5, c = (col>>3)&3.
// With this image BOTH access kinds the kernels need are bank-conflict-free (MI355X guide,
// section LDS): a ds_read_b128 row-fragment read (16 lanes = 16 distinct rows mod 16, same
// chunk) touches 16 distinct 16-byte slots of the 256-byte bank row, and a
// ds_read_b64_tr_b16 transposed read (per 32-lane half: 4 consecutive rows x the 4 chunks of
// one panel) does too.
#pragma once
#include <type_traits>
#include "common.h"

namespace attn {

constexpr int HD = 96;

__device__ __forceinline__ int panel_byte(int rows, int row, int col8 /* col/8 */) {
  const int p = col8 >> 2, c = col8 & 3;
  return p * rows * 64 + row * 64 + 16 * (c ^ ((row >> 2) & 3));
}

// A-operand fragment of a 32-row block for k-step ks (16 columns): lane -> row (lane&31),
// 8 contiguous columns at 16*ks + 8*(lane>>5)
template <int ROWS>
__device__ __forceinline__ bf16x8_t row_frag(const unsigned char* tile, int row0, int ks, int lane) {
  const int row = row0 + (lane & 31);
  return *(const bf16x8_t*)(tile + panel_byte(ROWS, row, 2 * ks + (lane >> 5)));
}

// Transposed fragment: operand element j of lane-half hh is tile[row = rbase + 8*(j>>2) + 4*hh +
// (j&3)][col = 32*panel + (lane&31)]  (rbase multiple of 16).  This is exactly the k-order in
// which a 32x32 f32 accumulator block, converted in place to bf16, presents its ROW index to
// the next MFMA (cdna guide: "An accumulator tile as the next MFMA's operand").
template <int ROWS>
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* tile, int rbase, int panel,
                                            int lane) {
  const int hh = lane >> 5, cg = (lane >> 4) & 1, i = lane & 15, q = i >> 2, pp = i & 3;
  const int r0 = rbase + 4 * hh;           // rows r0..r0+3, then r0+8..r0+11
  const int ch = 2 * cg + (pp >> 1);
  const int sw0 = (r0 >> 2) & 3, sw1 = ((r0 + 8) >> 2) & 3;
  const unsigned char* base = tile + panel * ROWS * 64 + 8 * (pp & 1);
  const s16x4_t lo = lds_read_tr16(base + (r0 + q) * 64 + 16 * (ch ^ sw0));
  const s16x4_t hi = lds_read_tr16(base + (r0 + 8 + q) * 64 + 16 * (ch ^ sw1));
  return make_bf16x8(lo, hi);
}

// registers 8s..8s+7 of an accumulator block -> bf16 operand fragment of k-step s
__device__ __forceinline__ bf16x8_t acc_to_frag(const f32x16_t& a, int s) {
  bf16x8_t f;
  if (s == 0) {
    f[0] = (__bf16)a[0]; f[1] = (__bf16)a[1]; f[2] = (__bf16)a[2]; f[3] = (__bf16)a[3];
    f[4] = (__bf16)a[4]; f[5] = (__bf16)a[5]; f[6] = (__bf16)a[6]; f[7] = (__bf16)a[7];
  } else {
    f[0] = (__bf16)a[8]; f[1] = (__bf16)a[9]; f[2] = (__bf16)a[10]; f[3] = (__bf16)a[11];
    f[4] = (__bf16)a[12]; f[5] = (__bf16)a[13]; f[6] = (__bf16)a[14]; f[7] = (__bf16)a[15];
  }
  return f;
}

// Cooperative global -> registers -> LDS staging of a [ROWS][COLS] bf16 tile (row stride ld
// elements in HBM, rows >= valid_rows zero-filled) into the panel image.
template <int ROWS, int COLS, int NT>
struct TileStager {
  static constexpr int CH = COLS / 8;
  static constexpr int CHUNKS = ROWS * CH;
  static constexpr int PER = (CHUNKS + NT - 1) / NT;
  uint4 r[PER];
  __device__ __forceinline__ void load(const bf16_t* __restrict__ src, size_t ld, int valid_rows,
                                       int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NT;
      const int row = c / CH, cc = c % CH;
      r[i] = make_uint4(0, 0, 0, 0);
      if (c < CHUNKS && row < valid_rows) r[i] = *(const uint4*)(src + (size_t)row * ld + cc * 8);
    }
  }
  __device__ __forceinline__ void store(unsigned char* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NT;
      if (c < CHUNKS) *(uint4*)(tile + panel_byte(ROWS, c / CH, c % CH)) = r[i];
    }
  }
};

// Direct-to-LDS staging of a [ROWS][COLS] bf16 tile into the panel image with
// global_load_lds_dwordx4: one wave-instruction writes 1 KB = 16 rows of one panel, lane ->
// (row = lane>>2, slot = lane&3); the XOR swizzle goes on the per-lane SOURCE address (the LDS
// destination of an LDS-DMA is always wave-uniform base + lane*16).  Rows >= valid_rows re-read
// the last valid row (callers mask those rows arithmetically).  Every wave issues exactly
// PER_WAVE instructions (the counted s_waitcnt vmcnt needs one immediate for all waves); waves
// past the end repeat the last instruction (same bytes to the same place).
template <int ROWS, int COLS, int NWAVES>
struct GldsTile {
  static_assert(ROWS % 16 == 0 && COLS % 32 == 0, "panel image geometry");
  static constexpr int RG = ROWS / 16;                   // 16-row groups per panel
  static constexpr int INSTRS = RG * (COLS / 32);
  static constexpr int PER_WAVE = (INSTRS + NWAVES - 1) / NWAVES;
  // per-lane element offsets inside a tile are tile-invariant: computed once, so a full tile
  // costs one 64-bit add per DMA instruction instead of ~10 VALU of address arithmetic
  unsigned off[PER_WAVE];
  __device__ __forceinline__ void init(size_t ld, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      int n = wave + i * NWAVES;
      if (n >= INSTRS) n = INSTRS - 1;
      const int panel = n / RG, rg = n % RG;
      const int row = rg * 16 + (lane >> 2);
      const int ch = (lane & 3) ^ ((row >> 2) & 3);
      off[i] = (unsigned)(row * ld + panel * 32 + ch * 8);
    }
  }
  __device__ static __forceinline__ unsigned char* dst_of(unsigned char* tile, int wave, int i) {
    int n = wave + i * NWAVES;
    if (n >= INSTRS) n = INSTRS - 1;
    return tile + (n / RG) * ROWS * 64 + (n % RG) * 1024;
  }
  // all ROWS rows exist
  __device__ __forceinline__ void issue_full(const bf16_t* __restrict__ src, unsigned char* tile,
                                             int wave) const {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + off[i]),
          (__attribute__((address_space(3))) void*)dst_of(tile, wave, i), 16, 0, 0);
  }
  // ragged tile: rows >= valid_rows re-read the last valid row
  __device__ static __forceinline__ void issue(const bf16_t* __restrict__ src, size_t ld,
                                               int valid_rows, unsigned char* tile, int wave,
                                               int lane) {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      int n = wave + i * NWAVES;
      if (n >= INSTRS) n = INSTRS - 1;
      const int panel = n / RG, rg = n % RG;
      const int row = rg * 16 + (lane >> 2);
      const int ch = (lane & 3) ^ ((row >> 2) & 3);
      const int grow = min(row, valid_rows - 1);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + (size_t)grow * ld + panel * 32 + ch * 8),
          (__attribute__((address_space(3))) void*)dst_of(tile, wave, i), 16, 0, 0);
    }
  }
  __device__ __forceinline__ void issue_auto(const bf16_t* __restrict__ src, size_t ld,
                                             int valid_rows, unsigned char* tile, int wave,
                                             int lane) const {
    if (valid_rows >= ROWS) issue_full(src, tile, wave);
    else issue(src, ld, valid_rows, tile, wave, lane);
  }
};

// max of three in one VALU op (fmaxf() compiles to canonicalise + v_max pairs on MFMA outputs)
__device__ __forceinline__ float max3(float a, float b, float c3) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c3));
  return r;
}

// NP transposed fragments (panels 0..NP-1, same rows) fetched by ONE inline-asm statement that
// also waits for them.  Needed in kernels that keep LDS-DMA in flight: hipcc puts an
// s_waitcnt vmcnt(0) in front of every ds_read_tr *builtin* that follows a global_load_lds
// (it cannot prove the intrinsic does not read the bytes being DMA'd), which drains the
// pipeline each tile; an asm read is invisible to that pass.  Outputs are early-clobber and the
// lgkmcnt wait sits inside the statement, so no register is consumed before its data landed
// (cdna guide 5.7, form (i)).
template <int ROWS, int NP>
__device__ __forceinline__ void tr_frags_asm(const unsigned char* tile, int rbase, int lane,
                                             bf16x8_t (&out)[NP]) {
  static_assert(NP >= 3 && NP <= 5, "3..5 panels");
  const int hh = lane >> 5, cg = (lane >> 4) & 1, i = lane & 15, q = i >> 2, pp = i & 3;
  const int r0 = rbase + 4 * hh;
  const int ch = 2 * cg + (pp >> 1);
  const int sw0 = (r0 >> 2) & 3, sw1 = ((r0 + 8) >> 2) & 3;
  const unsigned base = (unsigned)(size_t)tile + 8 * (pp & 1);
  const unsigned a_lo = base + (r0 + q) * 64 + 16 * (ch ^ sw0);
  const unsigned a_hi = base + (r0 + 8 + q) * 64 + 16 * (ch ^ sw1);
  constexpr int PS = ROWS * 64;   // panel stride in bytes
  s16x4_t l0, h0, l1, h1, l2, h2, l3, h3, l4, h4;
  if constexpr (NP == 3) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %6\n\tds_read_b64_tr_b16 %1, %7\n\t"
        "ds_read_b64_tr_b16 %2, %6 offset:%8\n\tds_read_b64_tr_b16 %3, %7 offset:%8\n\t"
        "ds_read_b64_tr_b16 %4, %6 offset:%9\n\tds_read_b64_tr_b16 %5, %7 offset:%9\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2)
        : "v"(a_lo), "v"(a_hi), "i"(PS), "i"(2 * PS)
        : "memory");
  } else if constexpr (NP == 4) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\t"
        "ds_read_b64_tr_b16 %2, %8 offset:%10\n\tds_read_b64_tr_b16 %3, %9 offset:%10\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:%11\n\tds_read_b64_tr_b16 %5, %9 offset:%11\n\t"
        "ds_read_b64_tr_b16 %6, %8 offset:%12\n\tds_read_b64_tr_b16 %7, %9 offset:%12\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3)
        : "v"(a_lo), "v"(a_hi), "i"(PS), "i"(2 * PS), "i"(3 * PS)
        : "memory");
  } else {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %10\n\tds_read_b64_tr_b16 %1, %11\n\t"
        "ds_read_b64_tr_b16 %2, %10 offset:%12\n\tds_read_b64_tr_b16 %3, %11 offset:%12\n\t"
        "ds_read_b64_tr_b16 %4, %10 offset:%13\n\tds_read_b64_tr_b16 %5, %11 offset:%13\n\t"
        "ds_read_b64_tr_b16 %6, %10 offset:%14\n\tds_read_b64_tr_b16 %7, %11 offset:%14\n\t"
        "ds_read_b64_tr_b16 %8, %10 offset:%15\n\tds_read_b64_tr_b16 %9, %11 offset:%15\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3),
          "=&v"(l4), "=&v"(h4)
        : "v"(a_lo), "v"(a_hi), "i"(PS), "i"(2 * PS), "i"(3 * PS), "i"(4 * PS)
        : "memory");
  }
  __builtin_amdgcn_sched_barrier(0);   // keep the MFMAs below the in-statement wait (guide rule 18)
  out[0] = make_bf16x8(l0, h0);
  out[1] = make_bf16x8(l1, h1);
  out[2] = make_bf16x8(l2, h2);
  if constexpr (NP >= 4) out[3] = make_bf16x8(l3, h3);
  if constexpr (NP >= 5) out[4] = make_bf16x8(l4, h4);
}

// ---- explicitly pipelined LDS fragment reads (shared by the forward and backward kernels) -----
// Left to itself hipcc emits `ds_read_b128 ; s_waitcnt lgkmcnt(0) ; v_mfma` per k-step -- every
// MFMA waits out the full LDS latency of its own operand -- and, worse, puts an `s_waitcnt
// vmcnt(0)` in front of compiler-visible LDS reads that follow an LDS-DMA, which drains the K/V
// ring every tile.  The reads are therefore issued by hand through inline asm (invisible to the
// waitcnt pass), a few fragments AHEAD of the MFMAs that consume them, and released by a counted
// `s_waitcnt lgkmcnt(N)` that carries the fragment registers as in/out operands so that no consumer
// can be scheduled above it.  LDS operations return in order, so "at most N outstanding" =
// everything older than the last N has landed.
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
__device__ __forceinline__ void lds_read128f(f32x4_t& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "i"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read_tr(s16x4_t& d, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_release1(bf16x8_t& f) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_release2(s16x4_t& lo, s16x4_t& hi) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo), "+v"(hi) : "n"(N) : "memory");
}

// A stream of NR row fragments (one ds_read_b128 each) feeding NR MFMAs, read D fragments ahead
// through a ring of D + 1 register sets.  rd(Int<j>, frag&) issues read j, mm(Int<j>, frag) is MFMA
// j.  prologue() may be issued early (e.g. before a VALU section); nothing else may issue LDS
// operations between prologue() and the end of run() -- the waits count every LDS operation.
template <int NR, int D>
struct RowStream {
  static_assert(D >= 1 && D <= NR, "lookahead");
  bf16x8_t ring[D + 1];
  template <typename RD>
  __device__ __forceinline__ void prologue(RD&& rd) {
    static_for<0, D>([&](auto J) { rd(J, ring[decltype(J)::value % (D + 1)]); });
  }
  template <typename RD, typename MM>
  __device__ __forceinline__ void run(RD&& rd, MM&& mm) {
    static_for<0, NR>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (j + D < NR) rd(Int<j + D>{}, ring[(j + D) % (D + 1)]);
      lgkm_release1<(j + D < NR) ? D : NR - 1 - j>(ring[j % (D + 1)]);
      mm(J, ring[j % (D + 1)]);
      __builtin_amdgcn_sched_barrier(0);   // MFMA j stays here: it covers the flight of the reads behind it
    });
  }
};
// The same for transposed fragments (two ds_read_b64_tr_b16 each).
template <int NR, int D>
struct TrStream {
  static_assert(D >= 1 && D <= NR, "lookahead");
  s16x4_t lo[D + 1], hi[D + 1];
  template <typename RD>
  __device__ __forceinline__ void prologue(RD&& rd) {
    static_for<0, D>([&](auto J) {
      constexpr int j = decltype(J)::value;
      rd(J, lo[j % (D + 1)], hi[j % (D + 1)]);
    });
  }
  template <typename RD, typename MM>
  __device__ __forceinline__ void run(RD&& rd, MM&& mm) {
    static_for<0, NR>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (j + D < NR) rd(Int<j + D>{}, lo[(j + D) % (D + 1)], hi[(j + D) % (D + 1)]);
      lgkm_release2<2 * ((j + D < NR) ? D : NR - 1 - j)>(lo[j % (D + 1)], hi[j % (D + 1)]);
      mm(J, make_bf16x8(lo[j % (D + 1)], hi[j % (D + 1)]));
      __builtin_amdgcn_sched_barrier(0);
    });
  }
};

// LDS-DMA of a [ROWS][COLS] bf16 tile into the panel image through a BUFFER descriptor: the
// per-lane byte offset of every piece is tile-invariant (a VGPR computed once), the tile's base
// is a scalar offset, so a piece costs `s_mov m0` + `buffer_load_dwordx4 ... lds` and no vector
// ALU work at all.  Only for tiles whose ROWS rows all exist: the scalar offset takes no part in
// the descriptor's range check, so rows past the end of the matrix would read whatever follows it.
// A ragged last tile goes through issue_clamped(): per-lane offsets with the row clamped to the
// last valid one (callers mask those rows arithmetically).
template <int ROWS, int COLS, int NWAVES>
struct BufTile {
  static_assert(ROWS % 16 == 0 && COLS % 32 == 0, "panel image geometry");
  static constexpr int RG = ROWS / 16, INSTRS = RG * (COLS / 32);
  static constexpr int PER_WAVE = (INSTRS + NWAVES - 1) / NWAVES;
  unsigned voff[PER_WAVE];
  __device__ __forceinline__ void init(int ld, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      int n = wave + i * NWAVES;
      if (n >= INSTRS) n = INSTRS - 1;
      const int panel = n / RG, rg = n % RG;
      const int row = rg * 16 + (lane >> 2);
      const int ch = (lane & 3) ^ ((row >> 2) & 3);
      voff[i] = (unsigned)(row * ld + panel * 32 + ch * 8) * 2u;
    }
  }
#if __HIP_DEVICE_COMPILE__
  template <typename RSRC>
  __device__ __forceinline__ void issue(RSRC rsrc, unsigned soff_bytes, unsigned char* tile, int wave) const {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      int n = wave + i * NWAVES;
      if (n >= INSTRS) n = INSTRS - 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrc, (__attribute__((address_space(3))) void*)(tile + (n / RG) * ROWS * 64 + (n % RG) * 1024),
          16, voff[i], soff_bytes, 0, 0);
    }
  }
  template <typename RSRC>
  __device__ static __forceinline__ void issue_clamped(RSRC rsrc, unsigned soff_bytes, int ld, int valid_rows,
                                                       unsigned char* tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      int n = wave + i * NWAVES;
      if (n >= INSTRS) n = INSTRS - 1;
      const int panel = n / RG, rg = n % RG;
      const int row = rg * 16 + (lane >> 2);
      const int ch = (lane & 3) ^ ((row >> 2) & 3);
      const int grow = min(row, valid_rows - 1);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrc, (__attribute__((address_space(3))) void*)(tile + (n / RG) * ROWS * 64 + (n % RG) * 1024),
          16, (unsigned)(grow * ld + panel * 32 + ch * 8) * 2u, soff_bytes, 0, 0);
    }
  }
  template <typename RSRC>
  __device__ __forceinline__ void issue_auto(RSRC rsrc, unsigned soff_bytes, int ld, int valid_rows,
                                             unsigned char* tile, int wave, int lane) const {
    if (valid_rows >= ROWS) issue(rsrc, soff_bytes, tile, wave);
    else issue_clamped(rsrc, soff_bytes, ld, valid_rows, tile, wave, lane);
  }
#endif
};

// A wave's own 32 operand rows (COLS bf16 columns of each, row stride ld elements in HBM) fetched COALESCED by
// LDS-DMA into a wave-private LDS region and read back as MFMA row fragments.  The row-per-lane register loads
// this replaces (lane -> its own row, 16 bytes per k-step) touch 32 different 128-byte lines per instruction and
// come back to every line once per k-step pair: by ablation (tools/diag/run_attn_ablate.sh, round 4) the eight
// such loads of the forward's Q rows cost 1.9 us of a 20.7 us launch at the 14x14 stage.  Here one 1-KiB piece
// = RPP whole rows (lane -> row lane / CPR, 16-byte chunk lane % CPR; the lanes left over re-read chunk 0 into
// the piece's pad), i.e. whole lines, each fetched once.  Inside a row the chunks are ROTATED by the row index
// (slot = (chunk + row) % CPR, applied on the source side -- the LDS side of an LDS-DMA is always base + 16 lane),
// so that the 16 lanes of a ds_read_b128 group, which read the same chunk of 16 different rows, hit different
// 16-byte slots of the 256-byte bank row.
template <int COLS>
struct RowStage {
  static constexpr int CPR = COLS / 8;                 // 16-byte chunks per row
  static constexpr int RPP = 64 / CPR;                 // rows per piece: 4 (128 columns), 3 (160), 5 (96)
  static constexpr int PIECES = (32 + RPP - 1) / RPP;
  static constexpr int BYTES = PIECES * 1024;          // LDS bytes per wave
  // base = row 0 of the matrix, first = this wave's first row, rows >= nrows re-read the last one
  __device__ static __forceinline__ void issue(const bf16_t* __restrict__ base, size_t ld, int first, int nrows,
                                               unsigned char* region, int lane) {
    int rl = lane / CPR, c = lane % CPR;
    if (rl >= RPP) { rl = RPP - 1; c = 0; }            // (left-over lanes: a harmless re-read into the pad)
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const int r = min(p * RPP + rl, 31);
      const int g = (c + CPR - r % CPR) % CPR;         // the chunk that belongs in slot c of row r
      const int row = min(first + r, nrows - 1);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(base + (size_t)row * ld + g * 8),
          (__attribute__((address_space(3))) void*)(region + p * 1024), 16, 0, 0);
    }
  }
  // fragment of k-step ks (columns 16 ks + 8 (lane >> 5) ..) of row lane & 31
  __device__ static __forceinline__ bf16x8_t frag(const unsigned char* region, int ks, int lane) {
    const int r = lane & 31, g = 2 * ks + (lane >> 5);
    return *(const bf16x8_t*)(region + (r / RPP) * 1024 + (r % RPP) * (CPR * 16) + ((g + r) % CPR) * 16);
  }
};

// Static wave priority for the tile loops (MI355X guide, "Two waves per SIMD" items 2 and 4): two waves share a SIMD --
// the two halves of an 8-wave workgroup, or one wave of each of two co-resident 4-wave workgroups -- and at equal
// priority the older one wins every VALU arbitration.  One s_setprio before the loop, never flipped.
// Diagnostic builds pick the rule with -DSVIT_ATTN_PRIO=<n> (tools/diag/build_variant.py); 0 = no priority.
#ifndef SVIT_ATTN_PRIO
#define SVIT_ATTN_PRIO 0
#endif
__device__ __forceinline__ void static_prio(int lin_wg, int wave, int nwaves) {
  bool hi = false;
  if (nwaves == 8) hi = (SVIT_ATTN_PRIO & 8) ? false : wave >= 4;   // the younger half of an 8-wave workgroup
  else if ((SVIT_ATTN_PRIO & 7) == 1) hi = lin_wg & 1;               // workgroup parity
  else if ((SVIT_ATTN_PRIO & 7) == 2) hi = (lin_wg >> 3) & 1;        // parity inside the XCD's dispatch order
  else if ((SVIT_ATTN_PRIO & 7) == 3) hi = (lin_wg >> 8) & 1;        // every second round of 256 workgroups
  if (SVIT_ATTN_PRIO != 0 && hi) __builtin_amdgcn_s_setprio(1);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace attn


// Workgroups are dealt to the 8 XCDs round-robin by linear id, and every XCD has its own L2.
// Remap the linear id so that CONSECUTIVE logical ids land on ONE XCD: all query tiles of a
// (batch, head) then share that XCD's L2 copy of K/V instead of fetching it up to 8 times over
// the fabric (rocprofv3 FETCH_SIZE of attn_fwd at 14x14: 77 MB against 20 MB of operands).
__device__ __forceinline__ int xcd_remap(int lin, int nwg) {
  const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
  return (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
}
