// HAOG losses of the image ranks, forward and unit gradients in ONE launch (SURVEY.md 8(f)
// rank 2; slowfast/models/losses.py:50-93,138-155, slowfast/utils/box_ops.py:10-77) -- gfx950.
//
// The reference selects the non-empty target rows with boolean indexing (`pred[mask]`,
// `if tar_mask.sum() > 0`): every step syncs the host twice and the row count is dynamic, which
// a replayed HIP graph cannot hold.  Here the masks stay arithmetic: one workgroup walks the
// few hundred rows, reduces (valid count, BCE, L1, 1-GIoU, contact CE) through LDS, and in the
// same launch writes d(loss_k)/d(pred) for each of the four losses -- no host round trip, fixed
// shapes, capturable.  `svit_haog_loss_bwd` scales them by the upstream gradients.
#include "common.h"
#include "../../include/svit_hip.h"

namespace {
constexpr int NT = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();                       // red[] may still be read from the previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// torch.minimum / torch.maximum backward: the winner takes the gradient, a tie splits it
__device__ __forceinline__ float w_lt(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

struct GiouOut { float giou, dcx, dcy, dw, dh; };

// diag(generalized_box_iou(xyxy(a), xyxy(b))) and its gradient w.r.t. a = (cx, cy, w, h)
__device__ GiouOut giou_cxcywh(const float a[4], const float b[4]) {
  const float ax0 = a[0] - 0.5f * a[2], ay0 = a[1] - 0.5f * a[3];
  const float ax1 = a[0] + 0.5f * a[2], ay1 = a[1] + 0.5f * a[3];
  const float bx0 = b[0] - 0.5f * b[2], by0 = b[1] - 0.5f * b[3];
  const float bx1 = b[0] + 0.5f * b[2], by1 = b[1] + 0.5f * b[3];
  const float aw = ax1 - ax0, ah = ay1 - ay0;
  const float area_a = aw * ah, area_b = (bx1 - bx0) * (by1 - by0);
  const float iw_raw = fminf(ax1, bx1) - fmaxf(ax0, bx0), ih_raw = fminf(ay1, by1) - fmaxf(ay0, by0);
  const float iw = fmaxf(iw_raw, 0.f), ih = fmaxf(ih_raw, 0.f);
  const float inter = iw * ih, uni = area_a + area_b - inter;
  const float cw_raw = fmaxf(ax1, bx1) - fminf(ax0, bx0), ch_raw = fmaxf(ay1, by1) - fminf(ay0, by0);
  const float cw = fmaxf(cw_raw, 0.f), ch = fmaxf(ch_raw, 0.f);
  const float area_c = cw * ch;
  GiouOut o;
  o.giou = inter / uni - (area_c - uni) / area_c;
  // giou = inter/uni - 1 + uni/area_c
  const float g_uni = -inter / (uni * uni) + 1.f / area_c;
  const float g_inter = 1.f / uni - g_uni;          // direct, and through uni = .. - inter
  const float g_c = -uni / (area_c * area_c);
  const float g_iw = iw_raw >= 0.f ? g_inter * ih : 0.f, g_ih = ih_raw >= 0.f ? g_inter * iw : 0.f;
  const float g_cw = cw_raw >= 0.f ? g_c * ch : 0.f, g_ch = ch_raw >= 0.f ? g_c * cw : 0.f;
  const float dx1 = g_iw * w_lt(ax1, bx1) + g_cw * w_lt(bx1, ax1) + g_uni * ah;
  const float dx0 = -g_iw * w_lt(bx0, ax0) - g_cw * w_lt(ax0, bx0) - g_uni * ah;
  const float dy1 = g_ih * w_lt(ay1, by1) + g_ch * w_lt(by1, ay1) + g_uni * aw;
  const float dy0 = -g_ih * w_lt(by0, ay0) - g_ch * w_lt(ay0, by0) - g_uni * aw;
  o.dcx = dx0 + dx1;
  o.dcy = dy0 + dy1;
  o.dw = 0.5f * (dx1 - dx0);
  o.dh = 0.5f * (dy1 - dy0);
  return o;
}

__global__ __launch_bounds__(NT) void haog_loss_kernel(
    const float* __restrict__ pred,      // [R,5]  (objectness logit, cx, cy, w, h)
    const float* __restrict__ tar,       // [R,4]  cxcywh, all-zero row = no object
    const float* __restrict__ contact,   // [Rc,5] logits
    const int64_t* __restrict__ ctar,    // [Rc]   class, < 0 = ignore
    float* __restrict__ losses,          // [8]: l1, bce, giou, contact CE, #boxes, #contacts, #bad
    float* __restrict__ g_l1,            // [R,4]  d l1 / d pred[:,1:]
    float* __restrict__ g_bce,           // [R]    d bce / d pred[:,0]
    float* __restrict__ g_giou,          // [R,4]
    float* __restrict__ g_contact,       // [Rc,5]
    int R, int Rc) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float n = 0.f, s_bce = 0.f, s_l1 = 0.f, s_giou = 0.f;
  for (int r = tid; r < R; r += NT) {
    float p[5], t[4];
#pragma unroll
    for (int i = 0; i < 5; ++i) p[i] = pred[(size_t)r * 5 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = tar[(size_t)r * 4 + i];
    const bool valid = !(t[0] == 0.f && t[1] == 0.f && t[2] == 0.f && t[3] == 0.f);
    const float z = valid ? 1.f : 0.f, x = p[0];
    s_bce += fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));   // BCE-with-logits, stable form
    if (valid) {
      n += 1.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) s_l1 += fabsf(p[1 + i] - t[i]);
      s_giou += 1.f - giou_cxcywh(p + 1, t).giou;
    }
  }
  float m = 0.f, s_ce = 0.f, bad = 0.f;
  for (int r = tid; r < Rc; r += NT) {
    const int64_t c = ctar[r];
    if (c < 0) continue;
    if (c > 4) { bad += 1.f; continue; }
    float l[5], mx = -INFINITY, se = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) { l[i] = contact[(size_t)r * 5 + i]; mx = fmaxf(mx, l[i]); }
#pragma unroll
    for (int i = 0; i < 5; ++i) se += expf(l[i] - mx);
    s_ce += mx + logf(se) - l[(int)c];
    m += 1.f;
  }
  n = block_sum(n, red);
  s_bce = block_sum(s_bce, red);
  s_l1 = block_sum(s_l1, red);
  s_giou = block_sum(s_giou, red);
  m = block_sum(m, red);
  s_ce = block_sum(s_ce, red);
  bad = block_sum(bad, red);
  const float inv_n = n > 0.f ? 1.f / n : 0.f, inv_m = m > 0.f ? 1.f / m : 0.f;
  const float inv_r = 1.f / (float)R;
  if (tid == 0) {
    losses[0] = s_l1 * inv_n * 0.25f;   // F.l1_loss(mean) over n x 4 entries; 0 when n == 0
    losses[1] = s_bce * inv_r;
    losses[2] = s_giou * inv_n;
    losses[3] = s_ce * inv_m;
    losses[4] = n;
    losses[5] = m;
    losses[6] = bad;
    losses[7] = 0.f;
  }
  // ---- unit gradients ------------------------------------------------------------------
  for (int r = tid; r < R; r += NT) {
    float p[5], t[4];
#pragma unroll
    for (int i = 0; i < 5; ++i) p[i] = pred[(size_t)r * 5 + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = tar[(size_t)r * 4 + i];
    const bool valid = !(t[0] == 0.f && t[1] == 0.f && t[2] == 0.f && t[3] == 0.f);
    g_bce[r] = (1.f / (1.f + expf(-p[0])) - (valid ? 1.f : 0.f)) * inv_r;
    float a[4] = {0.f, 0.f, 0.f, 0.f}, gg[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float d = p[1 + i] - t[i];
        a[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv_n * 0.25f;
      }
      const GiouOut o = giou_cxcywh(p + 1, t);
      gg[0] = -o.dcx * inv_n; gg[1] = -o.dcy * inv_n; gg[2] = -o.dw * inv_n; gg[3] = -o.dh * inv_n;
    }
    *(float4*)(g_l1 + (size_t)r * 4) = make_float4(a[0], a[1], a[2], a[3]);
    *(float4*)(g_giou + (size_t)r * 4) = make_float4(gg[0], gg[1], gg[2], gg[3]);
  }
  for (int r = tid; r < Rc; r += NT) {
    const int64_t c = ctar[r];
    float l[5], mx = -INFINITY, se = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) { l[i] = contact[(size_t)r * 5 + i]; mx = fmaxf(mx, l[i]); }
#pragma unroll
    for (int i = 0; i < 5; ++i) { l[i] = expf(l[i] - mx); se += l[i]; }
    const bool keep = c >= 0 && c <= 4;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      g_contact[(size_t)r * 5 + i] = keep ? (l[i] / se - (i == (int)c ? 1.f : 0.f)) * inv_m : 0.f;
  }
}

__global__ __launch_bounds__(NT) void haog_loss_bwd_kernel(
    const float* __restrict__ up,        // [4] upstream d(total)/d(l1, bce, giou, contact)
    const float* __restrict__ g_l1, const float* __restrict__ g_bce,
    const float* __restrict__ g_giou, const float* __restrict__ g_contact,
    float* __restrict__ dpred, float* __restrict__ dcontact, int R, int Rc) {
  const float u0 = up[0], u1 = up[1], u2 = up[2], u3 = up[3];
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i < R * 5) {
    const int r = i / 5, c = i % 5;
    dpred[i] = c == 0 ? u1 * g_bce[r] : u0 * g_l1[r * 4 + c - 1] + u2 * g_giou[r * 4 + c - 1];
  }
  if (i < Rc * 5) dcontact[i] = u3 * g_contact[i];
}
}  // namespace

extern "C" int svit_haog_loss(const float* pred, const float* tar, const float* contact,
                              const int64_t* contact_tar, float* losses, float* g_l1,
                              float* g_bce, float* g_giou, float* g_contact, int R, int Rc,
                              void* stream) {
  if (!pred || !tar || !contact || !contact_tar || !losses || !g_l1 || !g_bce || !g_giou ||
      !g_contact)
    return SVIT_ERR_ARG;
  if (R <= 0 || Rc <= 0 || R > (1 << 20) || Rc > (1 << 20)) return SVIT_ERR_SHAPE;
  if (((uintptr_t)g_l1 | (uintptr_t)g_giou) & 15) return SVIT_ERR_ALIGN;
  hipLaunchKernelGGL(haog_loss_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, pred, tar,
                     contact, contact_tar, losses, g_l1, g_bce, g_giou, g_contact, R, Rc);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_haog_loss_bwd(const float* upstream, const float* g_l1, const float* g_bce,
                                  const float* g_giou, const float* g_contact, float* dpred,
                                  float* dcontact, int R, int Rc, void* stream) {
  if (!upstream || !g_l1 || !g_bce || !g_giou || !g_contact || !dpred || !dcontact)
    return SVIT_ERR_ARG;
  if (R <= 0 || Rc <= 0 || R > (1 << 20) || Rc > (1 << 20)) return SVIT_ERR_SHAPE;
  const int n = 5 * (R > Rc ? R : Rc);
  hipLaunchKernelGGL(haog_loss_bwd_kernel, dim3((n + NT - 1) / NT), dim3(NT), 0,
                     (hipStream_t)stream, upstream, g_l1, g_bce, g_giou, g_contact, dpred,
                     dcontact, R, Rc);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Video-rank classification loss (round 6): nn.CrossEntropyLoss(reduction="mean") of VideoImageLoss
// (slowfast/models/losses.py:121,158) forward AND its unit gradient in ONE launch -- the replayed step spent four
// stock launches (log-softmax, NLL, and their backward kernels) plus autograd's fills on [B, 174] numbers.
//   loss = mean over the rows with label != -100 of (logsumexp(x) - x[label]);  dlogits = (softmax(x) - onehot) / #rows
// (ignore_index = -100 as torch's default; any other label outside [0, C) poisons the loss with NaN: fail loudly).
// One workgroup; a wave per row (rows wave, wave + 4, ...); the row losses meet in LDS and are added in row order.
namespace {
__global__ __launch_bounds__(256) void ce_loss_kernel(const float* __restrict__ x, const int64_t* __restrict__ lab, int B, int C,
                                                      float* __restrict__ loss, float* __restrict__ dx) {
  extern __shared__ float ce_rows[];      // [B] row losses, then [1] the count
  __shared__ int ce_cnt[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int cnt = 0;
  for (int r = wave; r < B; r += 4) {
    const float* xr = x + (size_t)r * C;
    const long y = lab[r];
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, xr[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(xr[c] - m);
    s = wave_sum(s);
    const bool ign = y == -100, bad = !ign && (y < 0 || y >= C);
    const float lse = m + __logf(s);
    if (lane == 0) {
      float v = 0.f;
      if (bad) v = __builtin_nanf("");
      else if (!ign) v = lse - xr[y];
      ce_rows[r] = v;
    }
    cnt += ign ? 0 : 1;
  }
  if (lane == 0) ce_cnt[wave] = cnt;
  __syncthreads();
  const int n = ce_cnt[0] + ce_cnt[1] + ce_cnt[2] + ce_cnt[3];
  const float inv = n > 0 ? 1.f / (float)n : __builtin_nanf("");       // (all rows ignored: NaN, as torch)
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int r = 0; r < B; ++r) t += ce_rows[r];
    *loss = t * inv;
  }
  for (int r = wave; r < B; r += 4) {
    const float* xr = x + (size_t)r * C;
    const long y = lab[r];
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, xr[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(xr[c] - m);
    s = wave_sum(s);
    const float is = inv / s;
    for (int c = lane; c < C; c += 64)
      dx[(size_t)r * C + c] = y == -100 ? 0.f : (__expf(xr[c] - m) * is - (c == y ? inv : 0.f));
  }
}
}  // namespace

extern "C" int svit_ce_loss(const float* logits, const int64_t* labels, int B, int C, float* loss, float* dlogits, void* stream) {
  if (!logits || !labels || !loss || !dlogits) return SVIT_ERR_ARG;
  if (B <= 0 || C <= 0 || B > 8192) return SVIT_ERR_SHAPE;
  hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(256), (size_t)B * sizeof(float), (hipStream_t)stream, logits, labels, B, C, loss, dlogits);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The step's random draws in ONE launch (round 6): the stochastic-depth factors of every block, floor(keep_b + U) / keep_b per
// (block, branch, sample) (slowfast/models/common.py:46-59), and the head's dropout factors, {0, 1 / (1 - p)} per element
// (nn.Dropout, slowfast/models/head_helper / video_model_builder.py head) -- five stock launches before (rand, add, floor,
// div, dropout).  Philox-4x32-10 keyed by `state[0]` (the seed), counter = (element / 4, draw number state[1]); the LAST
// workgroup to finish advances state[1], so a replayed HIP graph draws fresh numbers every replay with no host involvement
// (state lives in device memory).  Not torch's stream: the masks are as random, not the same numbers.
namespace {
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}
__device__ __forceinline__ void philox4(uint64_t seed, uint64_t draw, uint32_t idx, float (&u)[4]) {
  uint32_t c[4] = {idx, 0u, (uint32_t)draw, (uint32_t)(draw >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = (float)(c[i] >> 8) * (1.0f / 16777216.0f);      // [0, 1) on a 2^-24 grid
}
__global__ __launch_bounds__(256) void step_draws_kernel(unsigned long long* __restrict__ state, const float* __restrict__ keep,
                                                         int n_scale, int per_block, float* __restrict__ scales,
                                                         int n_drop, float p_drop, float* __restrict__ drop) {
  const uint64_t seed = state[0], draw = state[1];
  const int n4 = (n_scale + 3) / 4 + (n_drop + 3) / 4;
  const int s4 = (n_scale + 3) / 4;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += gridDim.x * blockDim.x) {
    float u[4];
    philox4(seed, draw, (uint32_t)q, u);
    if (q < s4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * q + e;
        if (i < n_scale) {
          const float k = keep[i / per_block];
          scales[i] = floorf(k + u[e]) / k;
        }
      }
    } else {
      const float on = 1.f / (1.f - p_drop);
      const int i0 = 4 * (q - s4);
      const float4 v = make_float4(u[0] >= p_drop ? on : 0.f, u[1] >= p_drop ? on : 0.f, u[2] >= p_drop ? on : 0.f, u[3] >= p_drop ? on : 0.f);
      if (i0 + 3 < n_drop && ((uintptr_t)drop & 15) == 0) {
        *(float4*)(drop + i0) = v;
      } else {
        const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (i0 + e < n_drop) drop[i0 + e] = ve[e];
      }
    }
  }
  // the last workgroup out advances the draw number and re-arms the ticket.  Every workgroup has READ the draw number before it
  // takes its ticket (the barrier below orders its threads' loads in front of the atomic), and the two plain stores only have to be
  // visible to the NEXT launch -- no device-scope fence (on a multi-XCD part a release fence writes the XCD's L2 back: ~15 us here)
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(&state[2], 1ull) == (unsigned long long)gridDim.x - 1) {
    state[1] = draw + 1;
    state[2] = 0;
  }
}
}  // namespace

extern "C" int svit_step_draws(uint64_t* state, const float* keep, int n_blocks, int per_block, float* scales,
                               int n_drop, float p_drop, float* drop, void* stream) {
  if (!state || n_blocks < 0 || per_block < 0 || n_drop < 0) return SVIT_ERR_ARG;
  const int n_scale = n_blocks * per_block;
  if ((n_scale > 0 && (!keep || !scales)) || (n_drop > 0 && !drop) || p_drop < 0.f || p_drop >= 1.f) return SVIT_ERR_ARG;
  const int n4 = (n_scale + 3) / 4 + (n_drop + 3) / 4;
  if (n4 == 0) return SVIT_OK;
  int blocks = (n4 + 255) / 256;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(step_draws_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)state, keep, n_scale,
                     per_block, scales, n_drop, p_drop, drop);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}
