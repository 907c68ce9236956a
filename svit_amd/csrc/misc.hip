// Memory-bound helpers: casts / transposes of the weight copies, DropPath backward scaling,
// im2col for the patch-embedding conv, special-token rows, max-pool skip path, and the
// optimiser tail (global grad norm + clip + AdamW) -- gfx950 only.
#include <mutex>
#include <unordered_map>
#include "common.h"
#include "../../include/svit_hip.h"

namespace {

__global__ void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      const float4 a = *(const float4*)(src + i), b = *(const float4*)(src + i + 4);
      uint4 o;
      o.x = pack_bf16x2(a.x, a.y); o.y = pack_bf16x2(a.z, a.w);
      o.z = pack_bf16x2(b.x, b.y); o.w = pack_bf16x2(b.z, b.w);
      *(uint4*)(dst + i) = o;
    } else {
      for (int64_t j = i; j < n; ++j) dst[j] = f32_to_bf16(src[j]);
    }
  }
}

// one 32x32 tile per block-iteration; table rows: {src_off, dst_off, R, C, ldd}: dst[c*ldd + r]
__global__ void transpose_cast_kernel(const float* __restrict__ src_base,
                                      bf16_t* __restrict__ dst_base,
                                      const int64_t* __restrict__ table, int n_mats) {
  __shared__ float tile[32][33];
  const int mat = blockIdx.y;
  if (mat >= n_mats) return;
  const int64_t so = table[mat * 5 + 0], dof = table[mat * 5 + 1];
  const int R = (int)table[mat * 5 + 2], C = (int)table[mat * 5 + 3];
  const int64_t ldd = table[mat * 5 + 4];
  const int tr = (R + 31) / 32, tc = (C + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
    const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
    for (int i = ty; i < 32; i += 8) {
      const int r = r0 + i, c = c0 + tx;
      tile[i][tx] = (r < R && c < C) ? src_base[so + (int64_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
      const int c = c0 + i, r = r0 + tx;  // dst[c][r]
      if (r < R && c < C) dst_base[dof + (int64_t)c * ldd + r] = f32_to_bf16(tile[tx][i]);
    }
    __syncthreads();
  }
}


// the same batched transposes from the bf16 mirror (round 4: the fp32 form reads 137 MB for 69 MB of output with
// 4-byte loads and 2-byte scattered stores; the mirror is written by the cast right before): 64 x 64 tiles, 16-byte
// loads along source rows, 16-byte stores along destination rows where the run is whole and aligned, through an
// LDS tile with odd-dword row pitch (the 8 two-byte column reads of a lane hit 8 different banks)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src_base,
                                                             bf16_t* __restrict__ dst_base,
                                                             const int64_t* __restrict__ table, int n_mats) {
  constexpr int PITCH = 66;                                // bf16 elements per tile row (33 dwords)
  __shared__ bf16_t tile[64 * PITCH];
  const int mat = blockIdx.y;
  if (mat >= n_mats) return;
  const int64_t so = table[mat * 5 + 0], dof = table[mat * 5 + 1];
  const int R = (int)table[mat * 5 + 2], C = (int)table[mat * 5 + 3];
  const int64_t ldd = table[mat * 5 + 4];
  const int tr = (R + 63) / 64, tc = (C + 63) / 64;
  const int tid = threadIdx.x, ch = tid & 7, rr = tid >> 3;           // 8 chunks of 8 elements x 32 rows
  const bool src_vec = ((so | C) & 7) == 0, dst_vec = ((dof | ldd) & 7) == 0;
  for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
    const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int r = r0 + rr + 32 * pass, c = c0 + ch * 8;
      alignas(16) bf16_t v[8];
      if (src_vec && r < R && c + 8 <= C) {
        *(uint4*)v = *(const uint4*)(src_base + so + (int64_t)r * C + c);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (r < R && c + e < C) ? src_base[so + (int64_t)r * C + c + e] : (bf16_t)0;
      }
      uint32_t* d = (uint32_t*)(tile + (rr + 32 * pass) * PITCH + ch * 8);    // 4-byte aligned (PITCH even)
      const uint32_t* w = (const uint32_t*)v;
      d[0] = w[0]; d[1] = w[1]; d[2] = w[2]; d[3] = w[3];
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int cl = rr + 32 * pass, c = c0 + cl, r = r0 + ch * 8;     // destination row c, elements r .. r+7
      alignas(16) bf16_t v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[(ch * 8 + e) * PITCH + cl];
      if (c < C) {
        bf16_t* o = dst_base + dof + (int64_t)c * ldd + r;
        if (dst_vec && r + 8 <= R) {
          *(uint4*)o = *(const uint4*)v;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (r + e < R) o[e] = v[e];
        }
      }
    }
    __syncthreads();
  }
}

__global__ void scale_cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                  const float* __restrict__ row_scale, int rows_per_sample,
                                  int64_t rows, int cols, int gather_L, int gather_N,
                                  int gather_off) {
  const int cpr = cols >> 2;  // float4 chunks per row
  const int64_t total = rows * cpr;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / cpr;
    const float s = row_scale ? row_scale[r / rows_per_sample] : 1.f;
    // optional row gather: output row r reads source row (r / L) * N + off + r % L
    const int64_t sr = gather_L > 0 ? (r / gather_L) * gather_N + gather_off + r % gather_L : r;
    const float4 v = ((const float4*)src)[sr * cpr + (i - r * cpr)];
    uint2 o;
    o.x = pack_bf16x2(v.x * s, v.y * s);
    o.y = pack_bf16x2(v.z * s, v.w * s);
    ((uint2*)dst)[i] = o;
  }
}

// dst(bf16)[r, 0..ldd) = [src(f32)[r, 0..C) | zeros]   (row-padded weight copies)
__global__ void pad_cast_rows_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int R,
                                     int C, int ldd) {
  const int64_t total = (int64_t)R * ldd;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ldd), c = (int)(i % ldd);
    dst[i] = f32_to_bf16(c < C ? src[(int64_t)r * C + c] : 0.f);
  }
}

// ---- patch embedding im2col: Conv3d(3->96, k(3,7,7), s(2,4,4), p(1,3,3)) -----------------
// One workgroup = one output row line (b, to, yo, all xo), walked in chunks of 62 output
// positions: the 3 x 3 x 7 input rows a chunk needs are fetched once with coalesced float4 loads
// into an LDS image (bf16; columns outside the frame are zero), then every 8-column chunk of the
// [Wo, 448] output is assembled from LDS and stored as one 16-byte vector.  (The direct version
// issued eight scattered 4-byte gathers per chunk.)
constexpr int I2C_XO = 62;                    // output positions per chunk
constexpr int I2C_COLS = I2C_XO * 4 + 4;      // image columns: x in [xc0*4 - 4, xc0*4 + 248)
constexpr int I2C_PITCH = I2C_COLS + 4;       // bf16 elements per image row
// (round 4: no division in either loop -- the load loop walks (row = wave + 4 k, float4 = lane), so a row's (c, kt, ky) is
// wave-uniform scalar arithmetic; in the store loop a thread keeps ONE 8-column chunk of the 448 for the whole workgroup, its
// eight (image row, kx) offsets in registers, and walks the output positions four apart: 95 -> ~60 us at 8 x 16 x 224^2)
__global__ __launch_bounds__(256) void im2col_patch_kernel(const float* __restrict__ video,
                                                           bf16_t* __restrict__ cols, int B, int T,
                                                           int H, int W, int To, int Ho, int Wo) {
  __shared__ bf16_t img[63 * I2C_PITCH];      // [(c*3+kt)*7+ky][x - x_start]
  const int yo = blockIdx.x % Ho, to = (blockIdx.x / Ho) % To, b = blockIdx.x / (Ho * To);
  const bool vec = !(W & 3) && !((uintptr_t)video & 15);
  bf16_t* out = cols + (((int64_t)b * To + to) * Ho + yo) * Wo * 448;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // store side: thread -> (position phase xq, chunk); offsets of the chunk's eight columns (col >= 441: the zero pad)
  const int xq = tid / 56, chunk = tid - xq * 56;
  int off8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int col = chunk * 8 + e;
    off8[e] = col < 441 ? (col / 7) * I2C_PITCH + 1 + col % 7 : -1;
  }
  for (int xc0 = 0; xc0 < Wo; xc0 += I2C_XO) {
    const int x_start = xc0 * 4 - 4;
    if (xc0) __syncthreads();                 // the previous chunk's readers are done
    if (vec) {
      for (int r = wave; r < 63; r += 4) {    // (wave-uniform row)
        const int ky = r % 7, kt = (r / 7) % 3, c = r / 21;
        const int t = to * 2 - 1 + kt, y = yo * 4 - 3 + ky, x = x_start + lane * 4;
        if (lane < I2C_COLS / 4) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (t >= 0 && t < T && y >= 0 && y < H && x >= 0 && x < W)
            v = *(const float4*)(video + (((int64_t)b * 3 + c) * T + t) * H * W + (int64_t)y * W + x);
          uint2 pk;
          pk.x = pack_bf16x2(v.x, v.y); pk.y = pack_bf16x2(v.z, v.w);
          *(uint2*)(img + r * I2C_PITCH + lane * 4) = pk;
        }
      }
    } else {                                  // odd widths / unaligned base: scalar loads
      for (int i = tid; i < 63 * I2C_COLS; i += 256) {
        const int r = i / I2C_COLS, xi = i % I2C_COLS;
        const int ky = r % 7, kt = (r / 7) % 3, c = r / 21;
        const int t = to * 2 - 1 + kt, y = yo * 4 - 3 + ky, x = x_start + xi;
        float v = 0.f;
        if (t >= 0 && t < T && y >= 0 && y < H && x >= 0 && x < W)
          v = video[(((int64_t)b * 3 + c) * T + t) * H * W + (int64_t)y * W + x];
        img[r * I2C_PITCH + xi] = f32_to_bf16(v);
      }
    }
    __syncthreads();
    const int n_xo = min(I2C_XO, Wo - xc0);
    if (xq < 4) {
      for (int xl = xq; xl < n_xo; xl += 4) {
        bf16_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = off8[e] >= 0 ? img[off8[e] + xl * 4] : (bf16_t)0;   // x = xo*4 - 3 + kx
        uint4 o;
        o.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16); o.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
        o.z = (uint32_t)v[4] | ((uint32_t)v[5] << 16); o.w = (uint32_t)v[6] | ((uint32_t)v[7] << 16);
        ((uint4*)out)[(size_t)(xc0 + xl) * 56 + chunk] = o;
      }
    }
  }
}

// x[b,0,:] = cls ; x[b,1+L+t*O+o,:] = objq[o] + (add_pos ? pos_t[t] : 0)
__global__ void fill_special_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                    const float* __restrict__ objq,
                                    const float* __restrict__ pos_t, int B, int N, int L, int Tx,
                                    int O, int C, int add_pos) {
  const int per = (1 + Tx * O) * C;
  const int64_t total = (int64_t)B * per;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per), r = (int)(i % per);
    const int tok = r / C, c = r % C;
    if (tok == 0) {
      x[((int64_t)b * N) * C + c] = cls[c];
    } else {
      const int t = (tok - 1) / O, o = (tok - 1) % O;
      x[((int64_t)b * N + 1 + L + (tok - 1)) * C + c] =
          objq[o * C + c] + (add_pos ? pos_t[t * C + c] : 0.f);
    }
  }
}

// backward of the above in one launch: g_cls[c] += sum_b dx[b,0,c]; g_obj[o,c] += sum_{b,t} dx[b,obj(t,o),c];
// g_pos[t,c] += sum_{b,o} dx[b,obj(t,o),c] (add_pos).  One thread per output element, B*Tx (resp. B*O) terms.
__global__ void special_grads_kernel(const float* __restrict__ dx, float* __restrict__ g_cls,
                                     float* __restrict__ g_obj, float* __restrict__ g_pos, int B, int N, int L,
                                     int Tx, int O, int C, int add_pos) {
  const int n_out = C + O * C + (add_pos ? Tx * C : 0);
  // EIGHT lanes per output element (round 5): lane part p sums terms p, p + 8, ... in batches of eight independent loads,
  // the eight partial sums meet in a fixed shuffle tree -- bit-reproducible, and the launch is two or three memory round
  // trips deep instead of sixteen (33 us -> a few: one thread per element walked 128 terms in sixteen dependent batches)
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = gid >> 3, part = gid & 7;
  const bool live = i < n_out;
  float acc = 0.f;
  auto sum_terms = [&](int n_terms, auto&& addr) {
    for (int j0 = part; j0 < n_terms; j0 += 64) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = j0 + 8 * e < n_terms ? dx[addr(j0 + 8 * e)] : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += v[e];
    }
  };
  int kind = 3, o = 0, c = 0, t = 0;
  if (live) {
    if (i < C) {
      kind = 0;
      sum_terms(B, [&](int b) { return (int64_t)b * N * C + i; });
    } else if (i < C + O * C) {
      kind = 1; o = (i - C) / C; c = (i - C) % C;
      sum_terms(B * Tx, [&](int j) { const int b = j / Tx, tt = j % Tx; return ((int64_t)b * N + 1 + L + tt * O + o) * C + c; });
    } else {
      kind = 2; t = (i - C - O * C) / C; c = (i - C - O * C) % C;
      sum_terms(B * O, [&](int j) { const int b = j / O, oo = j % O; return ((int64_t)b * N + 1 + L + t * O + oo) * C + c; });
    }
  }
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 4, 64);
  if (part == 0) {
    if (kind == 0) g_cls[i] += acc;
    else if (kind == 1) g_obj[o * C + c] += acc;
    else if (kind == 2) g_pos[t * C + c] += acc;
  }
}

// ---- max-pool skip: kernel (1,3,3), stride (1,2,2), pad (0,1,1) on patch tokens -----------
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                   uint8_t* __restrict__ idx, int B, int T, int H, int W, int Ho,
                                   int Wo, int n_obj, int C) {
  const int c4 = C >> 2;
  const int Nin = 1 + T * H * W + n_obj, Nout = 1 + T * Ho * Wo + n_obj;
  // (round 5: blockIdx.y = clip, 32-bit index arithmetic inside it -- the three 64-bit divisions per element of the flat
  //  index were a third of the kernel's instructions)
  const int b = blockIdx.y;
  const unsigned per = (unsigned)Nout * c4;
  for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < per; j += gridDim.x * blockDim.x) {
    const int tok = (int)(j / (unsigned)c4), cc = (int)(j - (unsigned)tok * c4);
    const int64_t i = (int64_t)b * per + j;
    const float4* xb = (const float4*)(x + (int64_t)b * Nin * C);
    float4 best;
    uchar4 bi = make_uchar4(255, 255, 255, 255);
    if (tok == 0) {
      best = xb[cc];
    } else if (tok > T * Ho * Wo) {
      best = xb[(int64_t)(1 + T * H * W + (tok - 1 - T * Ho * Wo)) * c4 + cc];
    } else {
      const int p = tok - 1, xo = p % Wo, yo = (p / Wo) % Ho, t = p / (Wo * Ho);
      best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      float4 v[9];                     // all nine taps in flight before the first compare
      bool ok[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int yy = yo * 2 - 1 + k / 3, xx = xo * 2 - 1 + k % 3;
        ok[k] = yy >= 0 && yy < H && xx >= 0 && xx < W;
        v[k] = xb[ok[k] ? (int64_t)(1 + (t * H + yy) * W + xx) * c4 + cc : (int64_t)cc];
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        if (!ok[k]) continue;
        const unsigned char tap = (unsigned char)k;
        if (v[k].x > best.x) { best.x = v[k].x; bi.x = tap; }
        if (v[k].y > best.y) { best.y = v[k].y; bi.y = tap; }
        if (v[k].z > best.z) { best.z = v[k].z; bi.z = tap; }
        if (v[k].w > best.w) { best.w = v[k].w; bi.w = tap; }
      }
    }
    ((float4*)y)[i] = best;
    ((uchar4*)idx)[i] = bi;
  }
}

// gather form: each input position sums the grads of the (<= 2x2) windows that selected it
// (DX16: dx as bf16 -- the operand of the dim-change projection's backward GEMMs, saves the cast pass)
template <bool DX16>
__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                   void* __restrict__ dx, int B, int T, int H, int W, int Ho,
                                   int Wo, int n_obj, int C) {
  const int c4 = C >> 2;
  const int Nin = 1 + T * H * W + n_obj, Nout = 1 + T * Ho * Wo + n_obj;
  const int b = blockIdx.y;
  const unsigned per = (unsigned)Nin * c4;
  for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < per; j += gridDim.x * blockDim.x) {
    const int tok = (int)(j / (unsigned)c4), cc = (int)(j - (unsigned)tok * c4);
    const int64_t i = (int64_t)b * per + j;
    const float4* db = (const float4*)(dy + (int64_t)b * Nout * C);
    const uchar4* ib = (const uchar4*)(idx + (int64_t)b * Nout * C);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tok == 0) {
      g = db[cc];
    } else if (tok > T * H * W) {
      g = db[(int64_t)(1 + T * Ho * Wo + (tok - 1 - T * H * W)) * c4 + cc];
    } else {
      const int p = tok - 1, xx = p % W, yy = (p / W) % H, t = p / (W * H);
      // the (<= 2 x 2) candidate windows: all four loads are issued before the first compare
      float4 d[4];
      uchar4 s[4];
      unsigned char tap[4];
      bool ok[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int yo = (yy >> 1) + (k >> 1), xo = (xx >> 1) + (k & 1);
        const int ky = yy - (yo * 2 - 1), kx = xx - (xo * 2 - 1);
        ok[k] = yo < Ho && xo < Wo && ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2;
        tap[k] = (unsigned char)(ky * 3 + kx);
        const int64_t o = ok[k] ? (int64_t)(1 + (t * Ho + yo) * Wo + xo) * c4 + cc : cc;
        d[k] = db[o];
        s[k] = ib[o];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!ok[k]) continue;
        if (s[k].x == tap[k]) g.x += d[k].x;
        if (s[k].y == tap[k]) g.y += d[k].y;
        if (s[k].z == tap[k]) g.z += d[k].z;
        if (s[k].w == tap[k]) g.w += d[k].w;
      }
    }
    if constexpr (DX16) {
      uint2 h;
      h.x = pack_bf16x2(g.x, g.y);
      h.y = pack_bf16x2(g.z, g.w);
      ((uint2*)dx)[i] = h;
    } else {
      ((float4*)dx)[i] = g;
    }
  }
}

// ---- second stage of the two-stage parameter-gradient reductions --------------------------
// One job = column sums of `partial` [nblocks][n] added into up to six destination vectors.  Up
// to SVIT_REDUCE_MAX_JOBS jobs share a launch (blockIdx.y = job): a transformer block's backward
// produces four of them (two LayerNorms, pooled-LN, conv wgrad), each latency-bound alone; since round 4 the host
// schedule keeps a group of four blocks' jobs in the queue (rotating scratch regions) and runs them as one launch.
constexpr int SVIT_REDUCE_MAX_JOBS = 16;     // (round 4: four transformer blocks' jobs per launch; 16 x 88 bytes of kernel argument)
struct SvitReduceJob {
  const float* partial;
  int nblocks, n;
  SvitReduceDst dst;
};
struct SvitReduceBatch {
  SvitReduceJob job[SVIT_REDUCE_MAX_JOBS];
  int count;
};

__global__ __launch_bounds__(1024) void svit_reduce_partials_kernel(SvitReduceBatch batch) {
  // block (32 columns x 32 row lanes): coalesced 128-byte row segments, 32x4 rows in flight per
  // iteration, so ~1000 partial rows are 8 iterations deep.  ONE workgroup owns a column block
  // from the first row to the last and adds in a fixed order: the result is bit-reproducible
  // (round 1 cut the rows into blockIdx.z chunks that met in fp32 atomics -- every LayerNorm /
  // pooling-conv parameter gradient then depended on the order the chunks happened to commit).
  __shared__ float red[32][33];
  const SvitReduceJob& j = batch.job[blockIdx.y];
  const float* __restrict__ partial = j.partial;
  const int n = j.n;
  if ((int)blockIdx.x * 32 >= n) return;
  const int r1 = j.nblocks;
  const int i = blockIdx.x * 32 + threadIdx.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int b = threadIdx.y;
    for (; b + 96 < r1; b += 128) {
      s0 += partial[(size_t)b * n + i];
      s1 += partial[(size_t)(b + 32) * n + i];
      s2 += partial[(size_t)(b + 64) * n + i];
      s3 += partial[(size_t)(b + 96) * n + i];
    }
    for (; b < r1; b += 32) s0 += partial[(size_t)b * n + i];
  }
  red[threadIdx.y][threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.y == 0 && i < n) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) s += red[k][threadIdx.x];
    int k = 0, lo = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q)
      if (i >= j.dst.end[q]) { k = q + 1; lo = j.dst.end[q]; }
    j.dst.ptr[k][i - lo] += s;
  }
}

// ---- optimiser tail ---------------------------------------------------------------------
// Deterministic two-stage sum of squares: replicas of a data-parallel job must compute the
// bit-identical clip coefficient from their (identical, all-reduced) gradients, so no atomics.
__global__ void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const float4 v = *(const float4*)(g + i);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (int64_t j = i; j < n; ++j) s += g[j] * g[j];
    }
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void sumsq_final_kernel(const float* __restrict__ partial, int nblocks,
                                   float* __restrict__ out) {
  // one wave, fixed order: lane l sums partial[l], partial[l+64], ... then a butterfly
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 64) s += partial[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) *out += s;
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                             float* __restrict__ m, float* __restrict__ v, int64_t n,
                             const float* __restrict__ sumsq, float max_norm, float lr, float b1,
                             float b2, float eps, float wd, float bc1, float bc2_sqrt,
                             float grad_scale) {
  float coef = grad_scale;
  if (sumsq && max_norm > 0.f) {
    const float total = sqrtf(*sumsq) * grad_scale;
    coef *= fminf(max_norm / (total + 1e-6f), 1.0f);
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * coef;
    float w = p[i] * (1.f - lr * wd);
    const float mi = m[i] * b1 + gi * (1.f - b1);
    const float vi = v[i] * b2 + gi * gi * (1.f - b2);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    w -= (lr / bc1) * (mi / denom);
    p[i] = w; m[i] = mi; v[i] = vi;
  }
}

inline unsigned grid_for(int64_t work_items, int block, unsigned cap = 8192) {
  int64_t b = (work_items + block - 1) / block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}
// out[r][c] = sum_j M[r][j] * T[j][c]  (c < 96): the rel-pos tables of a block at another resolution -- M is the cached
// [Lpad, h + w + t rows] matrix of resize blocks / identities (engine._rel), T the block's three adjacent fp32 tables.  fp32
// result (the backward's operand) and / or its bf16 copy (the forward's q . R^T operand) in ONE launch: the frames pass and
// the 312^2 eval path ran a library GEMM (13 us for ~2 MFLOP) plus a cast per block.
__global__ __launch_bounds__(128) void table_interp_kernel(const float* __restrict__ M, int J, const float* __restrict__ T,
                                                           float* __restrict__ out32, bf16_t* __restrict__ out16) {
  const int r = blockIdx.x, c = threadIdx.x;
  if (c >= 96) return;
  const float* m = M + (size_t)r * J;
  float acc = 0.f;
#pragma unroll 8
  for (int j = 0; j < J; ++j) acc += m[j] * T[(size_t)j * 96 + c];      // (m[j] is uniform: scalar loads; eight rows of T in flight)
  if (out32) out32[(size_t)r * 96 + c] = acc;
  if (out16) out16[(size_t)r * 96 + c] = f32_to_bf16(acc);
}
// every block of a forward pass in ONE launch (blockIdx.y = job; the jobs live in device memory: a cached table per geometry)
__global__ __launch_bounds__(128) void table_interp_batched_kernel(const svit_table_interp_job* __restrict__ jobs) {
  const svit_table_interp_job j = jobs[blockIdx.y];
  const int r = blockIdx.x, c = threadIdx.x;
  if (r >= j.rows || c >= 96) return;
  const float* m = j.M + (size_t)r * j.J;
  float acc = 0.f;
#pragma unroll 8
  for (int k = 0; k < j.J; ++k) acc += m[k] * j.tables[(size_t)k * 96 + c];
  if (j.out32) j.out32[(size_t)r * 96 + c] = acc;
  if (j.out16) ((bf16_t*)j.out16)[(size_t)r * 96 + c] = f32_to_bf16(acc);
}
}  // namespace

extern "C" int svit_version(void) { return 1; }

extern "C" int svit_table_interp_batched(const svit_table_interp_job* jobs_dev, int n_jobs, int max_rows, void* stream) {
  if (!jobs_dev || n_jobs <= 0 || n_jobs > 65535 || max_rows <= 0) return SVIT_ERR_ARG;
  hipLaunchKernelGGL(table_interp_batched_kernel, dim3(max_rows, n_jobs), dim3(128), 0, (hipStream_t)stream, jobs_dev);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_table_interp(const float* M, int rows, int J, const float* tables, float* out32, void* out16, void* stream) {
  if (!M || !tables || (!out32 && !out16) || rows <= 0 || J <= 0) return SVIT_ERR_ARG;
  hipLaunchKernelGGL(table_interp_kernel, dim3(rows), dim3(128), 0, (hipStream_t)stream, M, J, tables, out32, (bf16_t*)out16);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

// ---- the knob table (common.h: SvitKnob) -----------------------------------------------------------------------
static const int k_knob_default[SVIT_K_COUNT] = {
    /* NT_STAGES */ 0, /* NT_CFG */ -1, /* NT_BK */ 0, /* TN_STEP_US_X100 */ 85, /* TN_ATOMIC_TBS_X100 */ 75, /* TN_TILE */ 2,
    /* POOL_FWD */ 2, /* POOL_BWD */ 1, /* POOL_FWD_LARGE */ 1, /* ATTN_DKV_FORM */ 0, /* ATTN_FWD_SHORT */ 1, /* POOL_FRAME */ 2};
static int g_knob[SVIT_K_COUNT] = {0, -1, 0, 85, 75, 2, 2, 1, 1, 0, 1, 2};     // (= k_knob_default; accessed through __atomic builtins)
int svit_knob(int k) { return __atomic_load_n(&g_knob[k], __ATOMIC_RELAXED); }
int svit_knob_set(int k, int v) {
  if (k < 0 || k >= SVIT_K_COUNT) return SVIT_ERR_ARG;
  __atomic_store_n(&g_knob[k], v, __ATOMIC_RELAXED);
  return SVIT_OK;
}
void svit_knob_reset() {
  for (int k = 0; k < SVIT_K_COUNT; ++k) __atomic_store_n(&g_knob[k], k_knob_default[k], __ATOMIC_RELAXED);
}
// The declared diagnostics entry points (include/svit_hip.h).  Out-of-range values are refused, not clamped.
extern "C" int svit_debug_set(int key, int val) {
  if (key == 0) return (val == 0 || (val >= 2 && val <= 4)) ? svit_knob_set(SVIT_K_NT_STAGES, val) : SVIT_ERR_ARG;
  if (key == 1) return (val >= -1 && val <= 10) ? svit_knob_set(SVIT_K_NT_CFG, val) : SVIT_ERR_ARG;
  if (key == 2) return (val == 0 || val == 32 || val == 64) ? svit_knob_set(SVIT_K_NT_BK, val) : SVIT_ERR_ARG;
  return SVIT_ERR_ARG;
}
extern "C" int svit_debug_set_tn(int step_us_x100, int atomic_tbs_x100) {
  if (step_us_x100 > 0) svit_knob_set(SVIT_K_TN_STEP_US_X100, step_us_x100);
  if (atomic_tbs_x100 > 0) svit_knob_set(SVIT_K_TN_ATOMIC_TBS_X100, atomic_tbs_x100);
  return SVIT_OK;
}
extern "C" int svit_debug_set_tn_tile(int mode) {
  return (mode >= 0 && mode <= 3) ? svit_knob_set(SVIT_K_TN_TILE, mode) : SVIT_ERR_ARG;
}
extern "C" int svit_debug_set_pool(int key, int val) {
  if (key == 0) return (val >= 0 && val <= 3) ? svit_knob_set(SVIT_K_POOL_FWD, val) : SVIT_ERR_ARG;
  if (key == 1) return (val == 0 || val == 1) ? svit_knob_set(SVIT_K_POOL_BWD, val) : SVIT_ERR_ARG;
  if (key == 2) return (val == 0 || val == 1) ? svit_knob_set(SVIT_K_POOL_FWD_LARGE, val) : SVIT_ERR_ARG;
  if (key == 3) return (val >= 0 && val <= 2) ? svit_knob_set(SVIT_K_POOL_FRAME, val) : SVIT_ERR_ARG;
  return SVIT_ERR_ARG;
}
extern "C" int svit_attn_debug_set(int key, int val) {
  if (key == 0) return (val >= 0 && val <= 2) ? svit_knob_set(SVIT_K_ATTN_DKV_FORM, val) : SVIT_ERR_ARG;
  if (key == 3) return (val == 0 || val == 1) ? svit_knob_set(SVIT_K_ATTN_FWD_SHORT, val) : SVIT_ERR_ARG;
  return SVIT_ERR_ARG;
}
extern "C" int svit_debug_reset(void) { svit_knob_reset(); return SVIT_OK; }
extern "C" const char* svit_arch(void) { return "gfx950"; }

extern "C" int svit_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0) return SVIT_ERR_ARG;
  if (((uintptr_t)src | (uintptr_t)dst) & 15) return SVIT_ERR_ALIGN;
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for((n + 7) / 8, 256)), dim3(256), 0,
                     (hipStream_t)stream, src, (bf16_t*)dst, n);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_transpose_cast_batched(const float* src_base, void* dst_base,
                                           const int64_t* table, int n_mats, int max_tiles,
                                           void* stream) {
  if (!src_base || !dst_base || !table || n_mats <= 0) return SVIT_ERR_ARG;
  int gx = max_tiles < 1 ? 1 : (max_tiles > 256 ? 256 : max_tiles);
  hipLaunchKernelGGL(transpose_cast_kernel, dim3(gx, n_mats), dim3(256), 0, (hipStream_t)stream,
                     src_base, (bf16_t*)dst_base, table, n_mats);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}


extern "C" int svit_transpose_bf16_batched(const void* src_base, void* dst_base, const int64_t* table,
                                           int n_mats, int max_tiles, void* stream) {
  if (!src_base || !dst_base || !table || n_mats <= 0) return SVIT_ERR_ARG;
  if (((uintptr_t)src_base | (uintptr_t)dst_base) & 15) return SVIT_ERR_ALIGN;
  int gx = max_tiles < 1 ? 1 : (max_tiles > 256 ? 256 : max_tiles);
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3(gx, n_mats), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src_base, (bf16_t*)dst_base, table, n_mats);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_scale_cast(const float* src, void* dst, const float* row_scale,
                               int rows_per_sample, int64_t rows, int cols, int gather_L,
                               int gather_N, int gather_off, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || cols % 4 != 0) return SVIT_ERR_ARG;
  if (row_scale && rows_per_sample <= 0) return SVIT_ERR_ARG;
  if (gather_L < 0 || (gather_L > 0 && (gather_N < gather_L || gather_off < 0))) return SVIT_ERR_ARG;
  hipLaunchKernelGGL(scale_cast_kernel, dim3(grid_for(rows * (cols / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, src, (bf16_t*)dst, row_scale, rows_per_sample, rows,
                     cols, gather_L, gather_N, gather_off);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_pad_cast_rows(const float* src, void* dst, int R, int C, int ldd, void* stream) {
  if (!src || !dst || R <= 0 || C <= 0 || ldd < C) return SVIT_ERR_ARG;
  hipLaunchKernelGGL(pad_cast_rows_kernel, dim3(grid_for((int64_t)R * ldd, 256)), dim3(256), 0,
                     (hipStream_t)stream, src, (bf16_t*)dst, R, C, ldd);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_im2col_patch(const float* video, void* cols, int B, int T, int H, int W,
                                 void* stream) {
  if (!video || !cols || B <= 0 || T <= 0 || H <= 0 || W <= 0) return SVIT_ERR_ARG;
  const int To = (T + 2 - 3) / 2 + 1, Ho = (H + 6 - 7) / 4 + 1, Wo = (W + 6 - 7) / 4 + 1;
  hipLaunchKernelGGL(im2col_patch_kernel, dim3((unsigned)(B * To * Ho)), dim3(256), 0,
                     (hipStream_t)stream, video, (bf16_t*)cols, B, T, H, W, To, Ho, Wo);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_fill_special_tokens(float* x, const float* cls, const float* objq,
                                        const float* pos_t, int B, int N, int L, int Tx, int O,
                                        int C, int add_pos, void* stream) {
  if (!x || !cls || !objq || (add_pos && !pos_t)) return SVIT_ERR_ARG;
  if (N != 1 + L + Tx * O) return SVIT_ERR_SHAPE;
  const int64_t total = (int64_t)B * (1 + Tx * O) * C;
  hipLaunchKernelGGL(fill_special_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, cls, objq, pos_t, B, N, L, Tx, O, C, add_pos);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_special_token_grads(const float* dx, float* g_cls, float* g_obj, float* g_pos, int B, int N,
                                        int L, int Tx, int O, int C, int add_pos, void* stream) {
  if (!dx || !g_cls || !g_obj || (add_pos && !g_pos)) return SVIT_ERR_ARG;
  if (N != 1 + L + Tx * O || B <= 0 || C <= 0) return SVIT_ERR_SHAPE;
  const int n_out = C + O * C + (add_pos ? Tx * C : 0);
  hipLaunchKernelGGL(special_grads_kernel, dim3((n_out * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, dx, g_cls,
                     g_obj, g_pos, B, N, L, Tx, O, C, add_pos);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_maxpool_fwd(const float* x, float* y, uint8_t* idx, int B, int T, int H, int W,
                                int n_obj, int C, void* stream) {
  if (!x || !y || !idx || C % 4 != 0) return SVIT_ERR_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t per = (int64_t)(1 + T * Ho * Wo + n_obj) * (C / 4);
  if (per >= (1ll << 31) || B > 65535) return SVIT_ERR_SHAPE;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(per, 256, 2048), B), dim3(256), 0,
                     (hipStream_t)stream, x, y, idx, B, T, H, W, Ho, Wo, n_obj, C);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_maxpool_bwd(const float* dy, const uint8_t* idx, float* dx, int B, int T, int H,
                                int W, int n_obj, int C, void* stream) {
  if (!dy || !dx || !idx || C % 4 != 0) return SVIT_ERR_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t per = (int64_t)(1 + T * H * W + n_obj) * (C / 4);
  if (per >= (1ll << 31) || B > 65535) return SVIT_ERR_SHAPE;
  hipLaunchKernelGGL(maxpool_bwd_kernel<false>, dim3(grid_for(per, 256, 2048), B), dim3(256), 0,
                     (hipStream_t)stream, dy, idx, (void*)dx, B, T, H, W, Ho, Wo, n_obj, C);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}
extern "C" int svit_maxpool_bwd_bf16(const float* dy, const uint8_t* idx, void* dx_bf16, int B, int T, int H,
                                     int W, int n_obj, int C, void* stream) {
  if (!dy || !dx_bf16 || !idx || C % 4 != 0) return SVIT_ERR_ARG;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t per = (int64_t)(1 + T * H * W + n_obj) * (C / 4);
  if (per >= (1ll << 31) || B > 65535) return SVIT_ERR_SHAPE;
  hipLaunchKernelGGL(maxpool_bwd_kernel<true>, dim3(grid_for(per, 256, 2048), B), dim3(256), 0,
                     (hipStream_t)stream, dy, idx, dx_bf16, B, T, H, W, Ho, Wo, n_obj, C);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

// Deferred second-stage reductions, ONE QUEUE PER STREAM (host side, mutex-guarded): a stream
// enters deferred mode with svit_reduce_defer(1, stream); reduces launched on THAT stream are
// then queued and run as one launch -- on that same stream, i.e. behind the kernels that wrote
// their partial rows -- when it leaves the mode.  Work on any other stream (a side stream running
// weight-gradient kernels next to the chain) is never captured by a queue it did not open: its
// reduce runs at once on its own stream.  (Round 1 kept one process-global queue that ignored the
// launching stream; a reduce queued from a side stream was flushed on the main stream with no
// dependency on the kernel producing its rows.)
struct ReduceQueue {
  SvitReduceBatch batch;
  bool defer = false;
  ReduceQueue() { batch.count = 0; }
};
static std::mutex g_reduce_mu;
static std::unordered_map<hipStream_t, ReduceQueue> g_reduce_q;

static void reduce_launch(const SvitReduceBatch& b, hipStream_t st) {
  int max_n = 0;
  for (int i = 0; i < b.count; ++i)
    if (b.job[i].n > max_n) max_n = b.job[i].n;
  hipLaunchKernelGGL(svit_reduce_partials_kernel, dim3((max_n + 31) / 32, b.count), dim3(32, 32),
                     0, st, b);
}

void svit_launch_reduce(const float* partial, int nblocks, int n, SvitReduceDst dst, hipStream_t st) {
  SvitReduceJob job = {partial, nblocks, n, dst};
  SvitReduceBatch run;
  run.count = 0;
  {
    std::lock_guard<std::mutex> lk(g_reduce_mu);
    auto it = g_reduce_q.find(st);
    if (it != g_reduce_q.end() && it->second.defer) {
      ReduceQueue& q = it->second;
      if (q.batch.count == SVIT_REDUCE_MAX_JOBS) {      // queue full: run what we have
        run = q.batch;
        q.batch.count = 0;
      }
      q.batch.job[q.batch.count++] = job;
      job.partial = nullptr;                            // queued
    }
  }
  if (run.count) reduce_launch(run, st);
  if (job.partial) {
    SvitReduceBatch one;
    one.job[0] = job;
    one.count = 1;
    reduce_launch(one, st);
  }
}

// take the stream's queued jobs (and optionally set its mode) under the lock, launch outside it
static void reduce_take(hipStream_t st, int set_defer /* -1 = keep */, bool drop, SvitReduceBatch* out) {
  out->count = 0;
  std::lock_guard<std::mutex> lk(g_reduce_mu);
  auto it = g_reduce_q.find(st);
  if (it == g_reduce_q.end()) {
    if (set_defer != 1) return;
    it = g_reduce_q.emplace(st, ReduceQueue()).first;
  }
  ReduceQueue& q = it->second;
  if (!drop) *out = q.batch;
  q.batch.count = 0;
  if (set_defer >= 0) q.defer = set_defer != 0;
  if (!q.defer) g_reduce_q.erase(it);                  // streams come and go (graph capture)
}

extern "C" int svit_reduce_defer(int on, void* stream) {
  // entering deferred mode drops leftovers of an aborted block (their pointers may be stale);
  // leaving it runs the queue (see svit_reduce_flush)
  SvitReduceBatch run;
  reduce_take((hipStream_t)stream, on ? 1 : 0, on != 0, &run);
  if (run.count) reduce_launch(run, (hipStream_t)stream);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_reduce_flush(void* stream) {
  SvitReduceBatch run;
  reduce_take((hipStream_t)stream, -1, false, &run);
  if (run.count) reduce_launch(run, (hipStream_t)stream);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_reduce_reset(void* stream) {
  // error path of the host schedule: forget the stream's queued jobs and leave deferred mode
  SvitReduceBatch run;
  reduce_take((hipStream_t)stream, 0, true, &run);
  return SVIT_OK;
}

extern "C" int svit_sumsq(const float* g, int64_t n, float* sumsq, float* workspace,
                          int64_t workspace_floats, void* stream) {
  if (!g || !sumsq || !workspace || n <= 0) return SVIT_ERR_ARG;
  unsigned blocks = grid_for((n + 3) / 4, 256, 1024);
  if ((int64_t)blocks > workspace_floats) blocks = (unsigned)workspace_floats;
  if (blocks < 1) return SVIT_ERR_ARG;
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, workspace);
  SVIT_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace,
                     (int)blocks, sumsq);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_adamw_step(float* p, const float* g, float* m, float* v, int64_t n,
                               const float* sumsq, float max_norm, float lr, float beta1,
                               float beta2, float eps, float wd, int step, float grad_scale,
                               void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step < 1) return SVIT_ERR_ARG;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     p, g, m, v, n, sumsq, max_norm, lr, beta1, beta2, eps, wd, bc1, bc2s,
                     grad_scale);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}
