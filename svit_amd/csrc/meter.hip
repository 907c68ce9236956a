// Multi-view test ensemble on the device -- gfx950.
//
// The reference copies every batch of clip probabilities to the host and folds them into
// per-video sums one clip at a time in Python (slowfast/utils/meters.py:303-336, after
// tools/test_net.py:147-160), i.e. one device->host sync per iteration.  Here the per-video
// accumulators live in HBM and one launch folds a whole batch; a second launch at the end counts
// the top-k hits (slowfast/utils/metrics.py:9-50).  A few hundred floats per launch: neither HBM-
// nor MFMA-bound, the point is that the eval loop never waits for the host.
//
// Bit-exactness: fp32 addition is not associative, so the kernel keeps the reference's ORDER --
// the first clip of a video inside the batch (its "leader" workgroup) folds all of that video's
// clips of the batch in batch order; no atomics on the scores.
#include "common.h"
#include "../../include/svit_hip.h"

namespace {
constexpr int NT = 256;

__global__ __launch_bounds__(NT) void ensemble_update_kernel(
    const float* __restrict__ preds, const int64_t* __restrict__ labels,
    const int64_t* __restrict__ clip_ids, int N, int C, int num_clips, int num_videos, int mode,
    int repeat, float* __restrict__ video_preds, int64_t* __restrict__ video_labels,
    int64_t* __restrict__ clip_count, int* __restrict__ err) {
  const int n = blockIdx.x;
  const int64_t id = clip_ids[n];
  const int64_t vid = id / num_clips;
  if (id < 0 || vid >= num_videos) {
    if (threadIdx.x == 0) atomicAdd(err + 0, 1);     // clip id outside the table
    return;
  }
  for (int m = 0; m < n; ++m)                          // uniform over the block
    if (clip_ids[m] >= 0 && clip_ids[m] / num_clips == vid) return;   // not the leader
  // labels / counts: one lane, sequential like the reference's loop
  if (threadIdx.x == 0) {
    int64_t cur = video_labels[vid], cnt = 0;
    for (int m = n; m < N; ++m) {
      if (clip_ids[m] < 0 || clip_ids[m] / num_clips != vid) continue;
      if (cur > 0 && cur != labels[m]) atomicAdd(err + 1, 1);         // meters.py:318-322 assert
      cur = labels[m];
      ++cnt;
    }
    video_labels[vid] = cur;
    clip_count[vid] += cnt * repeat;
  }
  for (int c = threadIdx.x; c < C; c += NT) {
    float acc = video_preds[(size_t)vid * C + c];
    // `repeat`: the dataset lists every view NUM_ENSEMBLE_VIEWS times (identical frames in test
    // mode); the unique views are folded cyclically `repeat` times = the reference's order
    for (int rep = 0; rep < repeat; ++rep)
      for (int m = n; m < N; ++m) {
        if (clip_ids[m] < 0 || clip_ids[m] / num_clips != vid) continue;
        const float p = preds[(size_t)m * C + c];
        acc = mode == 0 ? acc + p : fmaxf(acc, p);
      }
    video_preds[(size_t)vid * C + c] = acc;
  }
}

__global__ __launch_bounds__(NT) void topk_correct_kernel(
    const float* __restrict__ video_preds, const int64_t* __restrict__ video_labels, int V, int C,
    const int* __restrict__ ks, int nk, int* __restrict__ counts, int* __restrict__ err) {
  __shared__ int red[NT / 64];
  const int v = blockIdx.x;
  const int64_t label = video_labels[v];
  if (label < 0 || label >= C) {
    if (threadIdx.x == 0) atomicAdd(err + 2, 1);
    return;
  }
  const float s = video_preds[(size_t)v * C + label];
  int above = 0;     // classes ranked before the label: larger score, or equal and lower index
  for (int c = threadIdx.x; c < C; c += NT) {
    const float p = video_preds[(size_t)v * C + c];
    above += (p > s) || (p == s && c < (int)label);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) above += __shfl_xor(above, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = above;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int rank = red[0] + red[1] + red[2] + red[3];
    for (int i = 0; i < nk; ++i)
      if (rank < ks[i]) atomicAdd(counts + i, 1);      // integer atomics: order-independent
  }
}
}  // namespace

extern "C" int svit_ensemble_update(const float* preds, const int64_t* labels,
                                    const int64_t* clip_ids, int N, int C, int num_clips,
                                    int num_videos, int mode, int repeat, float* video_preds,
                                    int64_t* video_labels, int64_t* clip_count, int* err,
                                    void* stream) {
  if (!preds || !labels || !clip_ids || !video_preds || !video_labels || !clip_count || !err)
    return SVIT_ERR_ARG;
  if (N <= 0 || N > 65535 || C <= 0 || num_clips <= 0 || num_videos <= 0 || repeat <= 0 ||
      (mode != 0 && mode != 1))
    return SVIT_ERR_SHAPE;
  hipLaunchKernelGGL(ensemble_update_kernel, dim3(N), dim3(NT), 0, (hipStream_t)stream, preds,
                     labels, clip_ids, N, C, num_clips, num_videos, mode, repeat, video_preds,
                     video_labels, clip_count, err);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}

extern "C" int svit_topk_correct(const float* video_preds, const int64_t* video_labels, int V,
                                 int C, const int* ks, int nk, int* counts, int* err,
                                 void* stream) {
  if (!video_preds || !video_labels || !ks || !counts || !err) return SVIT_ERR_ARG;
  if (V <= 0 || C <= 0 || nk <= 0 || nk > 8) return SVIT_ERR_SHAPE;
  hipLaunchKernelGGL(topk_correct_kernel, dim3(V), dim3(NT), 0, (hipStream_t)stream, video_preds,
                     video_labels, V, C, ks, nk, counts, err);
  SVIT_LAUNCH_CHECK();
  return SVIT_OK;
}
