// SViT head in two launches (slowfast/models/video_model_builder.py:408-551) -- gfx950.
//
// The head is O(B * 65 * 768) work: dropout, four tiny Linears on the cls / object rows of the final
// norm's output (class logits; box MLP + sigmoid; objectness logit; contact state of the first two
// objects of a frame), a concatenation.  As stock ATen ops under autograd that was ~45 launches of 5 us per
// training step (four hipBLASLt GEMMs that take 8-24 us each for < 1 MFLOP, slice / cat copies and their
// backward zero-fills, bias reductions, one AccumulateGrad add per parameter): 0.3 ms of a 13.5 ms step.
// Here: one forward launch (a wave per dot product), one backward launch (workgroup roles: parameter
// gradients accumulated straight into the flat gradient buffer; d(tokens) written whole -- zero rows for the
// patch tokens the head never read).  fp32 throughout, as the reference's head under autocast-off.
#include "common.h"
#include "../../include/svit_hip.h"

namespace {
constexpr int HEAD_NT = 256;

// x[b, r, c] = tokens[b, row(r), c] * keep[b, r, c]; r = 0 is the cls row, r >= 1 the objects (t-major)
__device__ __forceinline__ int head_row(const svit_head_args& a, int r) {
  return r == 0 ? 0 : a.N - a.T * a.O + (r - 1);
}

// out id -> (weight row pointer, bias, which output); ids: [0, n_cls) logits, then 4 box, 1 bce, 5 contact
struct HeadOut { const float* w; float b; int kind, k; };   // kind 0 logits, 1 box, 2 bce, 3 contact
__device__ __forceinline__ HeadOut head_out(const svit_head_args& a, int id) {
  HeadOut o;
  if (id < a.n_cls) { o.w = a.w_proj + (size_t)id * a.C; o.b = a.b_proj[id]; o.kind = 0; o.k = id; return o; }
  id -= a.n_cls;
  if (id < 4) { o.w = a.w_box + (size_t)id * a.C; o.b = a.b_box[id]; o.kind = 1; o.k = id; return o; }
  id -= 4;
  if (id < 1) { o.w = a.w_bce; o.b = a.b_bce[0]; o.kind = 2; o.k = 0; return o; }
  id -= 1;
  o.w = a.w_con + (size_t)id * a.C; o.b = a.b_con[id]; o.kind = 3; o.k = id;
  return o;
}

// one wave per (b, r, output): cls rows have n_cls outputs, object rows 5 (+5 for the first two of a frame)
__global__ __launch_bounds__(HEAD_NT) void head_fwd_kernel(svit_head_args a) {
  const int lane = threadIdx.x & 63;
  const int R = 1 + a.T * a.O;
  const int per_b = a.n_cls + a.T * a.O * 5 + a.T * 2 * 5;
  const int total = a.B * per_b;
  for (int w = blockIdx.x * (HEAD_NT / 64) + (threadIdx.x >> 6); w < total; w += gridDim.x * (HEAD_NT / 64)) {
    const int b = w / per_b;
    int id = w % per_b, r, out;
    if (id < a.n_cls) { r = 0; out = id; }
    else {
      id -= a.n_cls;
      if (id < a.T * a.O * 5) { r = 1 + id / 5; out = a.n_cls + id % 5; }
      else {
        id -= a.T * a.O * 5;
        const int t = id / 10, o = (id % 10) / 5;
        r = 1 + t * a.O + o; out = a.n_cls + 5 + id % 5;
      }
    }
    const HeadOut ho = head_out(a, out);
    const float* x = a.tokens + ((size_t)b * a.N + head_row(a, r)) * a.C;
    const float* kp = a.keep ? a.keep + ((size_t)b * R + r) * a.C : nullptr;
    float s = 0.f;
    for (int c = lane * 4; c < a.C; c += 256) {
      float4 xv = *(const float4*)(x + c);
      if (kp) { const float4 k = *(const float4*)(kp + c); xv.x *= k.x; xv.y *= k.y; xv.z *= k.z; xv.w *= k.w; }
      const float4 wv = *(const float4*)(ho.w + c);
      s += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
      // the object descriptors (the dropped features themselves) leave with the objectness output
      if (ho.kind == 2 && a.xobj) *(float4*)(a.xobj + ((size_t)b * (R - 1) + (r - 1)) * a.C + c) = xv;
    }
    s = wave_sum(s) + ho.b;
    if (lane == 0) {
      if (ho.kind == 0) a.logits[(size_t)b * a.n_cls + ho.k] = s;
      else if (ho.kind == 1) a.boxes[((size_t)b * (R - 1) + (r - 1)) * 5 + 1 + ho.k] = 1.f / (1.f + __expf(-s));
      else if (ho.kind == 2) a.boxes[((size_t)b * (R - 1) + (r - 1)) * 5] = s;
      else {
        const int t = (r - 1) / a.O, o = (r - 1) % a.O;
        a.contact[(((size_t)b * a.T + t) * 2 + o) * 5 + ho.k] = s;
      }
    }
  }
}

// d(pre-activation) of output `out` at (b, r); 0 where the output does not exist or has no gradient
__device__ __forceinline__ float head_dz(const svit_head_bwd_args& g, int b, int r, int kind, int k) {
  const svit_head_args& a = g.f;
  const int R = 1 + a.T * a.O;
  if (kind == 0) return (r == 0 && g.dlogits) ? g.dlogits[(size_t)b * a.n_cls + k] : 0.f;
  if (r == 0) return 0.f;
  const size_t ob = ((size_t)b * (R - 1) + (r - 1)) * 5;
  if (kind == 1) {
    if (!g.dboxes) return 0.f;
    const float s = a.boxes[ob + 1 + k];
    return g.dboxes[ob + 1 + k] * s * (1.f - s);
  }
  if (kind == 2) return g.dboxes ? g.dboxes[ob] : 0.f;
  const int t = (r - 1) / a.O, o = (r - 1) % a.O;
  if (o >= 2 || !g.dcontact) return 0.f;
  return g.dcontact[(((size_t)b * a.T + t) * 2 + o) * 5 + k];
}

// Backward roles by blockIdx.x (every sum runs over INDEPENDENT loads that the unrolled loops keep in flight;
// the first versions of this kernel chained 512 dependent round trips and took 90-180 us for < 1 MFLOP):
//   [0, n1)       class projection: block = (16 outputs, 256 channels); x of the B cls rows in registers,
//                 gw[o][c] += sum_b dz[b][o] x[b][c]; the block at channel 0 also adds the bias gradients
//   [n1, n1+n2)   box / objectness / contact weights: block = (32 object rows, 256 channels), every x read
//                 once for all ten outputs, partial sums leave through fp32 atomics (<= B T O / 32 adders per
//                 address); skipped entirely when neither dboxes nor dcontact is given (video ranks)
//   [.., +n3)     d(tokens) of the cls rows: block = 256 channels, every class-weight element read once for all B
//   rest          d(tokens) of all other rows: zeros for patch tokens, <= 10 terms + d(obj_desc) for objects
constexpr int HB_OC = 16, HB_RC = 32, HB_MAXB = 16, HB_CLS_MAX = 416;   // n_cls <= 416 (Kinetics-400 / SSv2-174)
__global__ __launch_bounds__(HEAD_NT) void head_bwd_kernel(svit_head_bwd_args g, int n1, int n2, int n3,
                                                           int rows_per_block) {
  __shared__ float dzs[HB_RC * 10 > HB_MAXB * HB_OC ? HB_RC * 10 : HB_MAXB * HB_OC];
  __shared__ float dzl[HB_MAXB * HB_CLS_MAX];       // whole dlogits rows of HB_MAXB clips (cls-row role)
  const svit_head_args& a = g.f;
  const int R = 1 + a.T * a.O, cchunks = (a.C + HEAD_NT - 1) / HEAD_NT;
  int blk = blockIdx.x;
  if (blk < n1) {
    const int oc = blk / cchunks, c = (blk % cchunks) * HEAD_NT + threadIdx.x, o0 = oc * HB_OC;
    const int no = min(HB_OC, a.n_cls - o0);
    float acc[HB_OC], sb = 0.f;
#pragma unroll
    for (int o = 0; o < HB_OC; ++o) acc[o] = 0.f;
    for (int b0 = 0; b0 < a.B; b0 += HB_MAXB) {         // (B = 63 stills on an image rank: four rounds)
      const int nb = min(HB_MAXB, a.B - b0);
      __syncthreads();
      for (int i = threadIdx.x; i < HB_MAXB * HB_OC; i += HEAD_NT) {
        const int b = i / HB_OC, o = i % HB_OC;
        dzs[i] = (b < nb && o < no && g.dlogits) ? g.dlogits[(size_t)(b0 + b) * a.n_cls + o0 + o] : 0.f;
      }
      __syncthreads();
      if (c < a.C) {
        float x[HB_MAXB];
#pragma unroll
        for (int b = 0; b < HB_MAXB; ++b) {
          x[b] = 0.f;
          if (b < nb) {
            x[b] = a.tokens[(size_t)(b0 + b) * a.N * a.C + c];
            if (a.keep) x[b] *= a.keep[(size_t)(b0 + b) * R * a.C + c];
          }
        }
#pragma unroll
        for (int o = 0; o < HB_OC; ++o)
#pragma unroll
          for (int b = 0; b < HB_MAXB; ++b) acc[o] += dzs[b * HB_OC + o] * x[b];
      }
      if ((int)threadIdx.x < no)
        for (int b = 0; b < nb; ++b) sb += dzs[b * HB_OC + threadIdx.x];
    }
    if (c < a.C)
#pragma unroll
      for (int o = 0; o < HB_OC; ++o)
        if (o < no) g.gw_proj[(size_t)(o0 + o) * a.C + c] += acc[o];
    if (blk % cchunks == 0 && (int)threadIdx.x < no) g.gb_proj[o0 + threadIdx.x] += sb;
    return;
  }
  blk -= n1;
  if (blk < n2) {
    if (!g.dboxes && !g.dcontact) return;
    const int rc = blk / cchunks, c = (blk % cchunks) * HEAD_NT + threadIdx.x;
    const int n_rows = a.B * (R - 1), i0 = rc * HB_RC, nr = min(HB_RC, n_rows - i0);
    for (int i = threadIdx.x; i < HB_RC * 10; i += HEAD_NT) {
      const int ri = i / 10, k = i % 10, row = i0 + ri;
      float dz = 0.f;
      if (ri < nr) {
        const int b = row / (R - 1), r = 1 + row % (R - 1);
        dz = k < 4 ? head_dz(g, b, r, 1, k) : k < 5 ? head_dz(g, b, r, 2, 0) : head_dz(g, b, r, 3, k - 5);
      }
      dzs[i] = dz;
    }
    __syncthreads();
    if (c < a.C) {
      float acc[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) acc[k] = 0.f;
#pragma unroll 8
      for (int ri = 0; ri < nr; ++ri) {
        const int row = i0 + ri, b = row / (R - 1), r = 1 + row % (R - 1);
        float xv = a.tokens[((size_t)b * a.N + head_row(a, r)) * a.C + c];
        if (a.keep) xv *= a.keep[((size_t)b * R + r) * a.C + c];
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[k] += dzs[ri * 10 + k] * xv;
      }
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        float* gw = k < 4 ? g.gw_box + (size_t)k * a.C : k < 5 ? g.gw_bce : g.gw_con + (size_t)(k - 5) * a.C;
        atomicAdd(gw + c, acc[k]);
      }
    }
    if (blk % cchunks == 0 && threadIdx.x < 10) {
      const int k = threadIdx.x;
      float sb = 0.f;
      for (int ri = 0; ri < nr; ++ri) sb += dzs[ri * 10 + k];
      atomicAdd(k < 4 ? g.gb_box + k : k < 5 ? g.gb_bce : g.gb_con + (k - 5), sb);
    }
    return;
  }
  blk -= n2;
  if (blk < n3) {
    // d(tokens[b, 0, :]) = keep * sum_o dlogits[b][o] W[o][:], HB_MAXB clips at a time; the clips' whole
    // dlogits rows are staged at once, so the class-weight loads of all rounds are independent of any barrier
    const int c = blk * HEAD_NT + threadIdx.x;
    const int ncp = (a.n_cls + HB_OC - 1) / HB_OC * HB_OC;           // <= HB_CLS_MAX (checked by the launcher)
    for (int b0 = 0; b0 < a.B; b0 += HB_MAXB) {
      const int nb = min(HB_MAXB, a.B - b0);
      __syncthreads();
      for (int i = threadIdx.x; i < HB_MAXB * ncp; i += HEAD_NT) {
        const int b = i / ncp, o = i % ncp;
        dzl[i] = (b < nb && o < a.n_cls && g.dlogits) ? g.dlogits[(size_t)(b0 + b) * a.n_cls + o] : 0.f;
      }
      __syncthreads();
      if (c < a.C) {
        float acc[HB_MAXB];
#pragma unroll
        for (int b = 0; b < HB_MAXB; ++b) acc[b] = 0.f;
        for (int o0 = 0; o0 < ncp; o0 += HB_OC) {
          float w[HB_OC];
#pragma unroll
          for (int o = 0; o < HB_OC; ++o) w[o] = o0 + o < a.n_cls ? a.w_proj[(size_t)(o0 + o) * a.C + c] : 0.f;
#pragma unroll
          for (int b = 0; b < HB_MAXB; ++b)
#pragma unroll
            for (int o = 0; o < HB_OC; ++o) acc[b] += dzl[b * ncp + o0 + o] * w[o];
        }
#pragma unroll
        for (int b = 0; b < HB_MAXB; ++b)
          if (b < nb) {
            float v = acc[b];
            if (a.keep) v *= a.keep[(size_t)(b0 + b) * R * a.C + c];
            g.dtokens[(size_t)(b0 + b) * a.N * a.C + c] = v;
          }
      }
    }
    return;
  }
  blk -= n3;
  // ---- every other row: zeros for the patch tokens, the object rows' few terms
  const int rb = (a.N - 1 + rows_per_block - 1) / rows_per_block;
  const int b = blk / rb, row0 = 1 + (blk % rb) * rows_per_block;
  for (int row = row0; row < min(a.N, row0 + rows_per_block); ++row) {
    const int r = row >= a.N - a.T * a.O ? 1 + row - (a.N - a.T * a.O) : -1;
    float* dx = g.dtokens + ((size_t)b * a.N + row) * a.C;
    float dz[10];
#pragma unroll
    for (int k = 0; k < 10; ++k)
      dz[k] = r < 0 ? 0.f : (k < 4 ? head_dz(g, b, r, 1, k) : k < 5 ? head_dz(g, b, r, 2, 0) : head_dz(g, b, r, 3, k - 5));
    for (int c = threadIdx.x * 4; c < a.C; c += HEAD_NT * 4) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r >= 1) {
#pragma unroll
        for (int k = 0; k < 10; ++k) {
          const float* w = k < 4 ? a.w_box + (size_t)k * a.C : k < 5 ? a.w_bce : a.w_con + (size_t)(k - 5) * a.C;
          const float4 wv = *(const float4*)(w + c);
          acc.x += dz[k] * wv.x; acc.y += dz[k] * wv.y; acc.z += dz[k] * wv.z; acc.w += dz[k] * wv.w;
        }
        if (g.dxobj) {
          const float4 d = *(const float4*)(g.dxobj + ((size_t)b * (R - 1) + (r - 1)) * a.C + c);
          acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
        }
        if (a.keep) {
          const float4 k4 = *(const float4*)(a.keep + ((size_t)b * R + r) * a.C + c);
          acc.x *= k4.x; acc.y *= k4.y; acc.z *= k4.z; acc.w *= k4.w;
        }
      }
      *(float4*)(dx + c) = acc;
    }
  }
}

int check_head(const svit_head_args& a) {
  if (!a.tokens || !a.w_proj || !a.b_proj || !a.w_box || !a.b_box || !a.w_bce || !a.b_bce || !a.w_con || !a.b_con ||
      !a.logits || !a.boxes || !a.contact)
    return SVIT_ERR_ARG;
  if (a.B <= 0 || a.T <= 0 || a.O < 2 || a.n_cls <= 0 || a.C <= 0 || a.C % 4 != 0 || a.N < 1 + a.T * a.O)
    return SVIT_ERR_SHAPE;
  if (((uintptr_t)a.tokens | (uintptr_t)a.w_proj | (uintptr_t)a.w_box | (uintptr_t)a.w_bce | (uintptr_t)a.w_con |
       (uintptr_t)a.keep | (uintptr_t)a.xobj) & 15)
    return SVIT_ERR_ALIGN;
  return SVIT_OK;
}
}  // namespace

extern "C" int svit_head_fwd(const svit_head_args* a, void* stream) {
  if (!a) return SVIT_ERR_ARG;
  if (int rc = check_head(*a)) return rc;
  const int total = a->B * (a->n_cls + a->T * a->O * 5 + a->T * 2 * 5);
  int blocks = (total + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)blocks), dim3(HEAD_NT), 0, (hipStream_t)stream, *a);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_head_bwd(const svit_head_bwd_args* g, void* stream) {
  if (!g) return SVIT_ERR_ARG;
  if (int rc = check_head(g->f)) return rc;
  if (!g->dtokens || !g->gw_proj || !g->gb_proj || !g->gw_box || !g->gb_box || !g->gw_bce || !g->gb_bce ||
      !g->gw_con || !g->gb_con)
    return SVIT_ERR_ARG;
  if (((uintptr_t)g->dtokens | (uintptr_t)g->dxobj) & 15) return SVIT_ERR_ALIGN;
  const svit_head_args& a = g->f;
  if (a.n_cls > HB_CLS_MAX) return SVIT_ERR_SHAPE;
  const int cchunks = (a.C + HEAD_NT - 1) / HEAD_NT;
  const int n1 = ((a.n_cls + HB_OC - 1) / HB_OC) * cchunks;
  const int n2 = ((a.B * a.T * a.O + HB_RC - 1) / HB_RC) * cchunks;
  const int n3 = cchunks;
  const int rows_per_block = 8, rb = (a.N - 1 + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)(n1 + n2 + n3 + a.B * rb)), dim3(HEAD_NT), 0, (hipStream_t)stream,
                     *g, n1, n2, n3, rows_per_block);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}
