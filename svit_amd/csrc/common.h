// Shared device helpers for the SViT HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits in HBM
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define SVIT_OK 0
#define SVIT_ERR_SHAPE (-2)
#define SVIT_ERR_ALIGN (-3)
#define SVIT_ERR_ARG (-4)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device), safe to call from any
// host thread (forward runs on the main thread, backward on autograd's device thread): one bit
// per device in an atomic mask; a lost race only repeats an idempotent call.  Replaces the
// process-global `static bool configured` flags (no global mutable state).
struct SvitOnce {
  unsigned long long mask;   // zero-initialised static storage; accessed through __atomic builtins
};
static inline int svit_max_lds_once(SvitOnce& once, const void* fn, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(&once.mask, __ATOMIC_ACQUIRE) & bit) return 0;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return (int)e;
  __atomic_fetch_or(&once.mask, bit, __ATOMIC_RELEASE);
  return 0;
}

// ---- tuning knobs (diagnostics) ----------------------------------------------------------------
// ONE process-wide table, defined in misc.hip.  It is written only through the svit_debug_* entry points
// declared in include/svit_hip.h ("diagnostics": tools/ and the variant-parity tests call them, the product path
// never does) and svit_debug_reset() restores every default.  The library reads NO environment variable: geometry
// constants that earlier rounds swept through getenv() are now compile-time macros (tools/diag/build_variant.py
// rebuilds one source with a different -D for an A/B).
enum SvitKnob {
  SVIT_K_NT_STAGES = 0,     // NT GEMM pipeline stages: 2..4, 0 = heuristic
  SVIT_K_NT_CFG,            // NT GEMM forced tile / ring configuration, -1 = heuristic, 8 = the pre-ring v2 heuristic
  SVIT_K_NT_BK,             // NT GEMM forced K-step: 32 / 64, 0 = heuristic
  SVIT_K_TN_STEP_US_X100,   // grouped TN planner: microseconds per k-step x 100
  SVIT_K_TN_ATOMIC_TBS_X100,  // grouped TN planner: TB/s of the fp32-atomic flush x 100
  SVIT_K_TN_TILE,           // grouped TN tile mode: 0 128x96 only, 1 isolated-launch heuristic, 2 128x192 everywhere, 3 128x192 where K % 192 == 0
  SVIT_K_POOL_FWD,          // small-plane pooling forward: 0 streaming, 1 VALU slab conv, 2 MFMA conv where ahead (default), 3 MFMA conv wherever it fits
  SVIT_K_POOL_BWD,          // small-plane pooling backward: 0 the three streaming launches, 1 (default) the fused plane-walk kernel where it fits
  SVIT_K_POOL_FWD_LARGE,    // large-plane pooling forward (blocks 0-3): 1 (default) staged conv + row-wise LayerNorm launch, 0 the streaming kernel
  SVIT_K_ATTN_DKV_FORM,     // attention dkv kernel: 0 heuristic, 1 four waves, 2 eight waves with query halves
  SVIT_K_ATTN_FWD_SHORT,    // attention forward T' = 1 tile for Nk <= 64: 1 on (default), 0 generic kernel
  SVIT_K_POOL_FRAME,        // one-plane volumes (T = 1): conv + LayerNorm from an LDS-staged plane -- 2 (default since round 6) every T = 1 pass, 1 no-grad passes only (frames pass), 0 never
  SVIT_K_COUNT
};
int svit_knob(int k);                 // misc.hip
int svit_knob_set(int k, int v);      // 0 or SVIT_ERR_ARG
void svit_knob_reset();

#define SVIT_LAUNCH_CHECK()                       \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even via the hardware convert (keeps NaN a NaN; see MI355X guide)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ float lo_bf16(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf16(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// raw v_exp_f32 (2^x; -inf -> 0).  exp2f() under hipcc expands to a ~5-instruction
// denormal-safe sequence per element, which made the attention kernels VALU-bound
// (16 VALU per MFMA measured); probabilities below 2^-126 may flush to 0, which is harmless.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// 32x32x16 bf16 MFMA: D = A(32x16) * B(16x32) + C.  Lane l: A[row l&31][k 8*(l>>5)+j],
// B[k 8*(l>>5)+j][col l&31]; C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
__device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-col block of 16-bit elements is
// delivered column-major; lane 4q+p supplies the address of row q, cols 4p..4p+3 and lane i
// receives column i (rows 0..3 in elements 0..3).  All 64 lanes must be active.
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
__device__ __forceinline__ s16x4_t lds_read_tr16(const void* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s16x4_t*)(__attribute__((address_space(3))) void*)p);
}

__device__ __forceinline__ bf16x8_t make_bf16x8(s16x4_t lo, s16x4_t hi) {
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// exact-erf GELU (nn.GELU() default, attention.py:481 / common.py:13) with erf evaluated by the
// Abramowitz-Stegun 7.1.26 rational (|abs err| <= 1.5e-7, far below bf16 resolution): one exp
// and one reciprocal instead of libm's branchy erff -- the fc1 / fc2-dgrad epilogues evaluate
// it 77M times per call in block 0.  exp(-x^2/2) is shared between the cdf and the pdf.
__device__ __forceinline__ void gelu_parts(float x, float* cdf, float* pdf_scaled) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float e = __expf(-ax * ax);                       // exp(-x^2/2)
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f +
                     t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;                  // erf(|x|/sqrt2)
  const float erfv = x < 0.f ? -erf_abs : erf_abs;
  *cdf = 0.5f * (1.0f + erfv);
  *pdf_scaled = 0.39894228040143268f * e;                 // standard normal pdf(x)
}
__device__ __forceinline__ float gelu_erf(float x) {
  float cdf, pdf;
  gelu_parts(x, &cdf, &pdf);
  return x * cdf;
}
__device__ __forceinline__ void gelu_fwd_grad(float x, float* y, float* dydx) {
#ifdef SVIT_DIAG_GELU_FREE      // (diagnostic build, TIMING ONLY: what the fc1 epilogue's GELU / GELU' arithmetic costs inside the step:
                                //  0.13-0.17 ms, profiles/r06_gelu_epilogue.txt; a Phi / phi table in LDS was measured 0.10 ms SLOWER than this math)
  *y = 0.5f * x; *dydx = 0.5f;
  return;
#endif
  float cdf, pdf;
  gelu_parts(x, &cdf, &pdf);
  *y = x * cdf;
  *dydx = cdf + x * pdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float cdf, pdf;
  gelu_parts(x, &cdf, &pdf);
  return cdf + x * pdf;
}

// Second stage of the two-stage parameter-gradient reductions: every block of the producing
// kernel stores its partial sums as one row of `partial` [nblocks][n] with plain stores, then
// this kernel adds the column sums into up to six destination vectors (segments of n).
// (Thousands of blocks atomically adding to the same few hundred addresses serialise at the
// memory side: MI355X guide, "Global float atomics", contention row.)
struct SvitReduceDst {
  float* ptr[6];
  int end[6];  // exclusive end of each segment within [0, n); unused tail entries = n
};
// Defined in misc.hip: launches the reduce now on `st`, or -- while `st` is in deferred mode
// (svit_reduce_defer(1, st) ... svit_reduce_defer(0, st)) -- queues it in THAT STREAM's queue so
// that all reduces of a transformer block run as ONE launch on the stream that produced their
// partial rows.  A launch on any other stream is never captured by the queue.
void svit_launch_reduce(const float* partial, int nblocks, int n, SvitReduceDst dst, hipStream_t st);

// Zero-fill as a kernel (16-byte stores): hipMemsetAsync turns into a memset node under stream
// capture, and those replay unreliably on ROCm 7.2 (svit_amd/graph.py needs every launch to be a
// plain kernel node).  bytes must be a multiple of 16 and p 16-byte aligned.
static __global__ void svit_zero_kernel(uint4* __restrict__ p, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
    p[i] = make_uint4(0u, 0u, 0u, 0u);
}
static inline void svit_launch_zero(void* p, size_t bytes, hipStream_t st) {
  const size_t n16 = bytes / 16;
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) return;
  hipLaunchKernelGGL(svit_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint4*)p, n16);
}

