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

}  // namespace attn
