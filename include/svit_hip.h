/* libsvit_hip.so -- C ABI of the MI355X-native SViT forward/backward kernels.
 *
 * Drop-in boundary: the reference has no native code; every entry point
 * below replaces an ATen op sequence of the reference's Python hot path (file:line cited
 * per function, relative to the reference tree).  The reference-side binding is the
 * `torch.autograd.Function` / ctypes stub shown in INTEGRATION.md.
 *
 * Conventions: all pointers are DEVICE pointers (hipMalloc'ed / torch CUDA tensors) unless
 * marked host; `stream` is a hipStream_t passed as void*; every function only enqueues work
 * on `stream` (no allocation, no synchronisation -> hipGraph-capturable) and returns 0 on
 * success, a negative SVIT_ERR_* for argument errors, or a positive hipError_t.
 * bf16 tensors are raw uint16 bit patterns.  Token order everywhere is
 * [cls | patches (t,y,x row-major) | objects]; head_dim is 96.
 */
#ifndef SVIT_HIP_H
#define SVIT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SVIT_HEAD_DIM 96

int svit_version(void);
const char* svit_arch(void); /* "gfx950" */

/* ---------------------------------------------------------------- GEMM family (K4) ---- */
/* nn.Linear everywhere: slowfast/models/attention.py:228,230,344-349,462,546-547,561;
 * slowfast/models/common.py:20-22,26-34; patch embed conv as implicit GEMM
 * (slowfast/models/stem_helper.py:309-320). */
enum {
  SVIT_EPI_BF16 = 0,   /* out(bf16)  = acc + bias                                   */
  SVIT_EPI_GELU = 1,   /* h = acc + bias; out(bf16) = gelu_erf(h); out2(bf16) = gelu_erf'(h) (out2 may be NULL) */
  SVIT_EPI_RESID = 2,  /* out(f32)   = aux(f32) + row_scale[row/rows_per_sample]*(acc+bias)
                          (proj / fc2 + DropPath + residual, attention.py:565,570)    */
  SVIT_EPI_F32 = 3,    /* out(f32)   = [out +] acc + bias, optional row remap         */
  SVIT_EPI_DGELU = 4,  /* out(bf16)  = acc * aux(bf16), aux = the saved gelu_erf'(h) (fc2 dgrad) */
  SVIT_EPI_RELQ = 5    /* rel-pos query side in ONE launch (round 3; replaces svit_gemm_nt + svit_relpos_gather,
                          attention.py:84-183): acc = q . Rcat^T is not stored; instead
                          relq_out[row, 96 + j] = bf16(bf16(acc[row, relq_map[(row % relq_rows) * relq_extra + j]])
                          * relq_scale) for j < relq_extra, 0 where the map holds -1 (cls / object rows, padding) */
};
typedef struct {
  const void* A; int32_t lda;      /* bf16 [M,K] row-major                             */
  const void* W; int32_t ldw;      /* bf16 [N,K] row-major (nn.Linear weight layout)   */
  const float* bias;               /* f32 [N] or NULL                                  */
  void* out; int32_t ldo;
  void* out2; int32_t ldo2;
  const void* aux; int32_t ldaux;
  const float* row_scale; int32_t rows_per_sample;
  int32_t M, N, K;                 /* N % 96 == 0, K % 32 == 0                         */
  int32_t epilogue; int32_t accumulate;
  int32_t remap_L, remap_N, remap_off; /* EPI_F32: out row = (r/L)*remap_N + remap_off + r%L */
  /* EPI_RELQ only (zero otherwise): */
  const int32_t* relq_map;         /* i32 [relq_rows, relq_extra]: column of acc for (token, j), or -1  */
  void* relq_out; int32_t relq_ld; /* bf16 qa rows [M, relq_ld]; columns 96 .. 96+relq_extra are written */
  int32_t relq_extra, relq_rows;   /* relq_extra = relq_ld - 96 in {32, 64}; relq_rows = tokens per (b, head) */
  float relq_scale;
} svit_gemm_args;
/* C[M,N] = A[M,K] * W[N,K]^T with fused epilogue (forward Linear; dgrad with W^T copy). */
int svit_gemm_nt(const svit_gemm_args* args, void* stream);
/* dW[N,K] (f32, atomically accumulated) += A[M,N]^T * B[M,K]  (Linear wgrad; split over M).
 * lda/ldb multiples of 8; K need not be (patch-embed wgrad: K = 441 inside ldb = 448). */
int svit_gemm_tn(const void* A, int lda, const void* B, int ldb, float* dW, int lddw,
                 int M, int N, int K, int splits, float* dbias /* f32 [N] += colsum(A), or NULL */,
                 void* stream);
/* Several independent wgrad GEMMs in ONE launch (the engine queues a block's Linear and rel-pos
 * weight gradients and flushes them together: the per-launch accumulator flush is amortised).
 * Same operand rules as svit_gemm_tn; any count (launched in groups of SVIT_TN_GROUP_MAX).
 * Replaces the per-layer autograd wgrads of slowfast/models/attention.py:377-409 (qkv, proj),
 * common.py:26-37 (fc1, fc2) and the rel-pos table grads of attention.py:77-139. */
#define SVIT_TN_GROUP_MAX 16
typedef struct {
  const void* A; const void* B; float* dW; float* dbias;   /* dbias may be NULL */
  int32_t lda, ldb, lddw, M, N, K;
} svit_tn_problem;
int svit_gemm_tn_grouped(const svit_tn_problem* probs, int count, void* stream);
/* ordered != 0: no split along the reduction rows -- one atomic add per dW element, i.e.
 * bit-reproducible weight gradients (regression-diff mode; slower). */
int svit_gemm_tn_grouped_ex(const svit_tn_problem* probs, int count, int ordered, void* stream);
/* dbias[N] (f32, atomically accumulated) += column sums of bf16 A[M,N]. */
int svit_colsum_bf16(const void* A, int lda, float* out, int M, int N, void* stream);

/* ------------------------------------------------------------- elementwise / casts ---- */
int svit_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
/* Rel-pos tables at another resolution (attention.py:84-137 interpolates rel_pos_h / _w / _t with F.interpolate when the
 * query / key grid differs from the table's): out[r][c] = sum_j M[r][j] * tables[j][c], c < 96, with M [rows, J] the
 * fp32 matrix of linear-interpolation weights (zero rows pad to a multiple of 96) and tables [J, 96] the block's three
 * adjacent fp32 tables.  out32 (fp32 [rows, 96]) and / or out16 (bf16 [rows, 96]) may be NULL (not both). */
int svit_table_interp(const float* M, int rows, int J, const float* tables, float* out32, void* out16, void* stream);
/* The same for every block of a forward pass in one launch: `jobs_dev` = n_jobs descriptors in DEVICE memory (the caller
 * caches the table per input geometry: the pointers are those of its persistent buffers), max_rows = the largest `rows`. */
typedef struct svit_table_interp_job {
  const float* M;        /* [rows, J] interpolation weights */
  const float* tables;   /* [J, 96] */
  float* out32;          /* [rows, 96] or NULL */
  void* out16;           /* bf16 [rows, 96] or NULL */
  int32_t rows, J;
} svit_table_interp_job;
int svit_table_interp_batched(const svit_table_interp_job* jobs_dev, int n_jobs, int max_rows, void* stream);
/* batched fp32 [R,C] -> bf16 [C,R] transposes described by a device table of
 * {src_off, dst_off, R, C, ldd} int64 quintuples: dst[c*ldd + r] (the W^T copies used by
 * dgrad; ldd > R places a table inside a wider row, e.g. the concatenated rel-pos tables). */
int svit_transpose_cast_batched(const float* src_base, void* dst_base, const int64_t* table,
                                int n_mats, int max_tiles, void* stream);
/* the same transposes, bf16 [R,C] -> bf16 [C,R], from the bf16 mirror of the weights (same table; offsets then index
 * the mirror): what the step uses -- half the bytes of the fp32 form, 16-byte accesses both ways.  max_tiles counts
 * 64 x 64 tiles here. */
int svit_transpose_bf16_batched(const void* src_base, void* dst_base, const int64_t* table,
                                int n_mats, int max_tiles, void* stream);
/* dst(bf16)[R,ldd] = [src(f32)[R,C] | 0]: row-padded bf16 copy (patch-embed weight 441 -> 448). */
int svit_pad_cast_rows(const float* src, void* dst, int R, int C, int ldd, void* stream);
/* dst(bf16)[r,:] = scale[r/rows_per_sample] * src(f32)[row(r),:]   (DropPath backward).
 * row(r) = r, or with gather_L > 0 the row gather (r / L) * gather_N + gather_off + r % L
 * (the patch rows [1, 1+L) of every sample of a [B, N, C] token tensor in one launch). */
int svit_scale_cast(const float* src, void* dst, const float* row_scale, int rows_per_sample,
                    int64_t rows, int cols, int gather_L, int gather_N, int gather_off,
                    void* stream);

/* ------------------------------------------- deferred second-stage reductions ---------- */
/* svit_layernorm_bwd, svit_pool_ln_bwd(_qkv) and svit_pool_conv_wgrad(_qkv) finish with a
 * "reduce partial rows" launch.  The queue is PER STREAM: between svit_reduce_defer(1, s) and
 * svit_reduce_flush(s) / svit_reduce_defer(0, s) the reduces launched ON STREAM s are queued
 * (host side) and run as ONE launch on s -- the engine brackets a transformer block's backward
 * with them.  Launches on any other stream are unaffected (they reduce at once on their own
 * stream).  While deferred, every producer must be given its own workspace region (the rows are
 * read at the flush), and the gradients are final only after the flush.  svit_reduce_defer(1)
 * drops leftovers of an aborted bracket; svit_reduce_reset(s) is the error path: forget s's queue
 * and leave deferred mode.  Thread-safe (mutex-guarded map keyed by stream; no other global
 * mutable state in the library). */
int svit_reduce_defer(int on, void* stream);
int svit_reduce_flush(void* stream);
int svit_reduce_reset(void* stream);

/* ---------------------------------------------------------------- LayerNorm (K3) ------ */
/* nn.LayerNorm(eps=1e-6): attention.py:501,531; video_model_builder.py:69,233. */
int svit_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16,
                       float* y_f32, float* mean, float* rstd, int64_t rows, int C, float eps,
                       void* stream);
/* dx = [dres +] LN'(dy); dgamma/dbeta += column sums (two-stage: per-block partial rows in
 * `workspace`, then one reduce launch -- no same-address atomics). dy is f32 or bf16.
 * dx_bf16 (optional): also bf16(row_scale[row / rows_per_sample] * dx) -- the DropPath-scaled
 * operand of the next backward GEMM (common.py:46-59), fused instead of a separate cast pass;
 * row_scale may be NULL (scale 1). */
int svit_layernorm_bwd(const void* dy /* f32, or bf16 when dy_is_bf16 */, int dy_is_bf16,
                       const float* x, const float* gamma, const float* mean,
                       const float* rstd, const float* dres, float* dx, void* dx_bf16,
                       const float* row_scale, int rows_per_sample, float* dgamma,
                       float* dbeta, int64_t rows, int C, float* workspace,
                       int64_t workspace_floats, void* stream);

/* ------------------------------------------------------------ patch embedding (K1/K2) - */
/* im2col for Conv3d(3->96, k(3,7,7), s(2,4,4), p(1,3,3)) (stem_helper.py:309-320):
 * video f32 [B,3,T,H,W] -> cols bf16 [B*T'*H'*W', 448] (441 taps, zero-padded to 448). */
int svit_im2col_patch(const float* video, void* cols, int B, int T, int H, int W, void* stream);
/* The same operand straight from decoded uint8 frames: frames u8
 * [V,T,Hs,Ws,3] (T H W C per video, `frames_bytes` = size of the buffer), lut bf16 [3,256] =
 * bf16((u/255 - mean[c]) / std[c]) (datasets/utils.py:287-303), crops int32 [B,3] = (source video,
 * y0, x0) per output clip (transform.py:327-342; NULL: clip b = video b at (0,0)); the S x S
 * window is normalised, zero-padded and laid out as cols bf16 [B*T'*S'*S', 448]. The crop table
 * is NOT range-checked on the device: the caller guarantees y0 + S <= Hs, x0 + S <= Ws, v < V. */
int svit_im2col_patch_u8(const uint8_t* frames, int64_t frames_bytes, const void* lut,
                         const int32_t* crops, void* cols, int B, int T, int Hs, int Ws, int S,
                         void* stream);
/* cls / object token rows of the block-0 input (video_model_builder.py:326-363). */
int svit_fill_special_tokens(float* x, const float* cls, const float* objq, const float* pos_t,
                             int B, int N, int L, int Tx, int O, int C, int add_pos, void* stream);
/* its backward in one launch: g_cls [C] += sum_b dx[b,0]; g_obj [O,C] += sum_{b,t} dx[b,1+L+t*O+o];
 * g_pos [Tx,C] += sum_{b,o} of the same rows (add_pos) -- dx f32 [B,N,C]. */
int svit_special_token_grads(const float* dx, float* g_cls, float* g_obj, float* g_pos, int B, int N,
                             int L, int Tx, int O, int C, int add_pos, void* stream);

/* ------------------------------------------------- pooled q/k/v (K5, K6, K3 fused) ---- */
/* attention_pool with depthwise Conv3d(96,96,3^3,stride (1,s,s),pad 1) + object gain +
 * LayerNorm(96) on every token (attention.py:13-65, 263-304).
 * in : qkv bf16 [B, N, 3, h, 96]  (which = 0/1/2 selects q/k/v)
 * out: bf16 [B, h, Nout, ld_out] columns 0..95 = LN(pooled); pre: bf16 [B,h,Nout,96]
 *      pre-LN pooled values (saved for backward); mean/rstd f32 [B*h*Nout]; pre, mean and rstd
 *      may all be NULL (no-grad passes: nothing is saved).
 * mode 1 (keys) additionally writes the one-hot key coordinates at columns
 *      96 + [y | kh + x | kh + kw + t] used by the in-MFMA relative-position bias. */
typedef struct {
  const void* qkv; int32_t which;
  const float* conv_w;            /* f32 [96, 27]                                      */
  const float* gamma; const float* beta;
  void* out; int32_t ld_out;
  void* pre; float* mean; float* rstd;
  int32_t B, heads, T, H, W, n_obj; /* input tokens N = 1 + T*H*W + n_obj               */
  int32_t stride_hw;               /* spatial stride s (temporal stride is 1)          */
  int32_t mode;                    /* 0 = plain (q, v), 1 = keys (+one-hot)            */
  float eps;
  float out_scale;                 /* columns 0..95 of out = LN(pooled) * out_scale (0 = 1): the keys
                                    * carry scale * log2(e) so that qa . ka^T is the score in the
                                    * log2 domain (svit_attn_fwd)                          */
  /* optional, the q tensor of the *_qkv entry points only (round 3; NULL = off): also write the rel-pos columns
   * out[:, 96 + j] = bf16(bf16(LN(q) . relq_R[relq_map[token, j]]) * relq_scale), 0 where the map is -1 --
   * what svit_gemm_nt with SVIT_EPI_RELQ does in a launch of its own.  relq_R bf16 [relq_lpad, 96] (the
   * concatenated tables, relq_lpad % 96 == 0), relq_map i32 [Nout, ld_out - 96].  The slab LayerNorm
   * kernel multiplies it on the matrix pipe from the rows it has just normalised; when the tensor takes
   * another path the entry point runs the GEMM itself, so the columns are written either way. */
  const void* relq_R; const int32_t* relq_map; int32_t relq_lpad; float relq_scale;
} svit_pool_args;
int svit_pool_ln_fwd(const svit_pool_args* a, void* stream);

/* backward, step 1: LayerNorm(96) backward per pooled token.
 * dout: up to three addends: d_main (bf16 or f32, row stride ld_main), d_res (bf16 ctx grad for
 * the residual-pooling path, rows [B, Nout, h*96], skipped for cls), d_extra (f32, or bf16 with extra_is_bf16, [..,96]).
 * writes dpre bf16 [B,h,Nout,96]; dgamma/dbeta accumulated (two-stage via workspace). */
typedef struct {
  const void* d_main; int32_t main_is_f32; int32_t ld_main;
  const void* d_res; const void* d_extra;
  const void* pre; const float* mean; const float* rstd; const float* gamma;
  void* dpre; float* dgamma; float* dbeta;
  int32_t B, heads, Nout;
  float* workspace; int64_t workspace_floats;   /* scratch for the two-stage dgamma/dbeta sum */
  int32_t main_parts; int64_t main_part_stride; /* f32 d_main only: sum of main_parts planes, main_part_stride
                                                 * floats apart (svit_attn_bwd's dk / dv); 0 / 1 = one plane */
  int32_t extra_is_bf16;                        /* d_extra holds bf16 (the rel-pos D . R^T GEMM's bf16 epilogue: half the bytes) */
} svit_pool_ln_bwd_args;
int svit_pool_ln_bwd(const svit_pool_ln_bwd_args* a, void* stream);
/* backward, step 2: depthwise-conv dgrad + cls/object rows -> dqkv bf16 [B,N,3,h,96] slice `which`;
 * step 3: depthwise-conv wgrad incl. the object-gain path; dw f32 [96,27] accumulated.
 * (argument blocks of svit_pool_conv_bwd_qkv below, one pair per tensor) */
typedef struct {
  const void* dpre; const float* conv_w; void* dqkv; int32_t which;
  int32_t B, heads, T, H, W, n_obj, stride_hw;
} svit_pool_dgrad_args;
typedef struct {
  const void* dpre; const void* qkv; int32_t which; float* dw;
  int32_t B, heads, T, H, W, n_obj, stride_hw;
  float* workspace; int64_t workspace_floats;
} svit_pool_wgrad_args;
/* The four pooling entry points for q, k and v of one block in ONE launch each (args[0..2] =
 * which 0, 1, 2; same qkv / B / heads / T / H / W / n_obj, individual strides and outputs).
 * This is what the engine calls: at the 14x14 and 7x7 stages each stencil is a short latency
 * chain, so three side by side cost the time of one (attention.py:263-304 runs them serially).
 * The two-stage reductions use args[0].workspace (>= 1024 * 3 * 27 * 96 floats for wgrad). */
int svit_pool_ln_fwd_qkv(const svit_pool_args* args3, void* stream);
/* Stride-1 stencils that keep their input in LDS.  svit_pool_weight_sel turns a list of depthwise weights
 * (fp32 [96][27] each, at src_base + src_off[i]) into "selector" tables dst[i][27][96] (uint32:
 * bf16(w) in the half of the dword that matches the channel's position in a packed bf16 pair); run
 * once per step for all blocks.  The *_sel entry point takes the three tables of a block.  Who reads
 * them: the SLAB forward (round 3; planes of <= 196 tokens, i.e. the 14x14 and 7x7 stages) fetches a
 * channel group's 27 x 24 entries as SCALAR operands (s_load) -- its stencil has no vector weight loads
 * at all; the LDS-tiled kernels of round 2 (56x56 planes; halo ring in LDS) build their own LDS weight
 * image from conv_w and only consult WHETHER a table is given.  Without tables every tensor runs the
 * streaming kernels. */
int svit_pool_weight_sel(const float* src_base, const int64_t* src_off, uint32_t* dst, int n_tables,
                         void* stream);
int svit_pool_ln_fwd_qkv_sel(const svit_pool_args* args3, const uint32_t* const* sel3, void* stream);
int svit_pool_ln_bwd_qkv(const svit_pool_ln_bwd_args* args3, void* stream);
#ifdef SVIT_DIAG_POOL_STREAMING
/* DIAGNOSTIC BUILD ONLY (round 6: csrc/pool.hip compiled with -DSVIT_DIAG_POOL_STREAMING, tools/diag/build_variant.py): the
 * streaming conv backward of rounds 1-4 -- gather-form dgrad and LDS-tiled wgrad, per tensor and for q, k, v in one launch
 * each.  The product library does not export them: svit_pool_conv_bwd_qkv takes every block of every configuration the suite
 * runs (tools/diag/pool_bwd_paths.py) and fails loudly where its plan does not fit. */
int svit_pool_conv_dgrad(const svit_pool_dgrad_args* a, void* stream);
int svit_pool_conv_wgrad(const svit_pool_wgrad_args* a, void* stream);
int svit_pool_conv_dgrad_qkv(const svit_pool_dgrad_args* args3, void* stream);
int svit_pool_conv_wgrad_qkv(const svit_pool_wgrad_args* args3, void* stream);
#endif
/* Steps 2 + 3 together (what the engine calls; round 5): conv dgrad AND conv wgrad of q, k, v in ONE launch for every
 * stride and plane of the model (csrc/pool.hip::pool_bwd_fused_kernel: a workgroup stages the dpre halo of its chunk of
 * planes / rows once in LDS and walks its input tokens once, producing dqkv and per-workgroup partial rows of the three dw
 * that the second-stage reduce sums in a fixed order).  Since round 6 the ONLY conv backward of the product library: where
 * not even one unit row of three padded planes fits LDS, the workspace is smaller than the plan's partial rows, or with
 * svit_debug_set_pool(1, 0), it returns SVIT_ERR_SHAPE (a diagnostic build falls through to the streaming launches).
 * dgrad3[i] / wgrad3[i] describe the same `which` = i (same dpre, stride, dims); wgrad3[0].workspace >=
 * B * heads * chunks * 3 * 27 * 96 floats. */
int svit_pool_conv_bwd_qkv(const svit_pool_dgrad_args* dgrad3, const svit_pool_wgrad_args* wgrad3,
                           void* stream);
/* floats of wgrad3[0].workspace that call needs for these tensors (B * heads * chunks of its plan * 3 * 27 * 96; grows with the
 * batch), -1 = the fused kernel has no plan for them, < -1 = error code.  Only the shape fields of dgrad3 are read. */
int64_t svit_pool_conv_bwd_workspace(const svit_pool_dgrad_args* dgrad3);

/* ------------------------------------------- decomposed rel-pos bias, query side (K9/K10) */
/* cal_rel_pos_spatial / cal_rel_pos_temporal (attention.py:84-183) restated as
 * relq[q, j] = q . R_j(q) with j over [k_h | k_w | k_t]; stored (divided by the softmax
 * scale) in columns 96.. of the augmented query so that the bias is added by the QK^T MFMA
 * against the one-hot key columns.  tables are f32 [rows,96] already resized to
 * 2*max(q,k)-1 rows; idx_* are int32 [q_n, k_n] row tables (dist.long()). */
typedef struct {
  void* qa; int32_t ld;           /* bf16 [B,h,Nq,ld]: reads cols 0..95, writes 96..96+J-1 */
  const float* rel_h; const float* rel_w; const float* rel_t;
  const int32_t* idx_h; const int32_t* idx_w; const int32_t* idx_t;
  int32_t B, heads, qt, qh, qw, kt, kh, kw, n_obj;
  float inv_scale;
} svit_relq_args;
int svit_relpos_q_fwd(const svit_relq_args* a, void* stream);
typedef struct {
  const void* qa; const void* dqa; int32_t ld;   /* bf16; dqa cols 96.. hold d(relq/scale)   */
  const float* rel_h; const float* rel_w; const float* rel_t;
  const int32_t* idx_h; const int32_t* idx_w; const int32_t* idx_t;
  float* dq_extra;                /* f32 [B,h,Nq,96] written (0 for cls/objects)          */
  float* drel_h; float* drel_w; float* drel_t; /* accumulated                           */
  int32_t rows_h, rows_w, rows_t;
  int32_t B, heads, qt, qh, qw, kt, kh, kw, n_obj;
  float inv_scale;
  float* workspace; int64_t workspace_floats;
} svit_relq_bwd_args;
int svit_relpos_q_bwd(const svit_relq_bwd_args* a, void* stream);
/* GEMM formulation of the forward (the one the engine uses): P[tokens, ldp] = q . Rcat^T with
 * svit_gemm_nt over the concatenated tables (rows row_h.., row_w.., row_t..), then this gather
 * writes qa[:, 96 + j] = P[token, row_x + idx] * inv_scale (zeros for cls / objects / j >= J). */
typedef struct {
  const void* P; int32_t ldp; void* qa; int32_t ld;
  const int32_t* idx_h; const int32_t* idx_w; const int32_t* idx_t;
  int32_t row_h, row_w, row_t;
  int32_t B, heads, qt, qh, qw, kt, kh, kw, n_obj;
  float inv_scale;
} svit_relq_gather_args;
int svit_relpos_gather(const svit_relq_gather_args* a, void* stream);
/* GEMM formulation of the same backward (the one the engine uses): D[tokens, ldd] (bf16,
 * zero-filled here) receives d(relq) at column off_{h,w,t} + idx; then
 * drel_x += D[:, sec_x]^T q (svit_gemm_tn) and dq_extra = D Rcat (svit_gemm_nt). */
typedef struct {
  const void* dqa; int32_t ld; void* D; int32_t ldd;
  const int32_t* idx_h; const int32_t* idx_w; const int32_t* idx_t;
  int32_t off_h, off_w, off_t;
  int32_t B, heads, qt, qh, qw, kt, kh, kw, n_obj;
  float inv_scale;
} svit_relq_scatter_args;
int svit_relpos_scatter(const svit_relq_scatter_args* a, void* stream);

/* ------------------------------------------------ fused pooled attention (K8-K12) ------ */
/* (q*scale)@k^T + rel-pos bias -> softmax -> @v -> + pooled q (all tokens but cls) ->
 * merge heads (attention.py:429-461).
 * qa bf16 [B,h,Nq,DA], ka bf16 [B,h,Nk,DA] (DA = 128 or 160), v bf16 [B,h,Nk,96],
 * ctx bf16 [B,Nq,h*96], lse2 f32 [B,h,Nq] (log2-domain log-sum-exp, saved for backward).
 * Operand convention (round 3): qa . ka^T IS the attention score in the log2 domain --
 *   ka[:, 0:96]  = scale * log2(e) * LN(pool(k))   (svit_pool_args.out_scale of the key tensor),
 *   ka[:, 96+j]  = one-hot key coordinates [y | kh + x | kh + kw + t],
 *   qa[:, 0:96]  = LN(pool(q)), qa[:, 96+j] = log2(e) * (q . R_j) (svit_relpos_gather's inv_scale
 *   = log2 e) -- so the kernels exponentiate with exp2 and no multiply.  `scale` is only used to
 *   express dk with respect to the UN-scaled pooled keys: dk = scale * sum_q P (dP - delta) q.
 * bias_cols: how many of the DA - 96 rel-pos columns carry data (kt + kh + kw); the kernel only
 * multiplies the 16-column k-steps that do.  0 = all of them. */
typedef struct {
  const void* qa; const void* ka; const void* v; void* ctx; float* lse2;
  int32_t B, heads, Nq, Nk, DA; float scale; int32_t bias_cols;
} svit_attn_fwd_args;
int svit_attn_fwd(const svit_attn_fwd_args* a, void* stream);
typedef struct {
  const void* qa; const void* ka; const void* v; const void* ctx; const void* dctx;
  const float* lse2; float* delta;  /* delta: f32 [B,h,2,Nq] scratch (-lse2/c and -rowsum(dO*O) planes) */
  void* dqa;                        /* bf16 [B,h,Nq,DA]                                  */
  float* dk; float* dv;             /* f32 [parts,B,h,Nk,96], OVERWRITTEN: parts = svit_attn_bwd_parts(args)
                                     * partial planes (one per chunk of the query range) whose SUM is the
                                     * gradient; svit_pool_ln_bwd adds them while it reads (main_parts) */
  int32_t B, heads, Nq, Nk, DA, q_splits; float scale;
  int32_t bias_cols;                /* as in svit_attn_fwd_args: kt + kh + kw, 0 = all DA - 96 */
  /* optional (round 3; NULL / 0 = off): the dq kernel also writes the scattered matrix of the rel-pos backward,
   * relD bf16 [B*h*Nq, relD_ld] = zeros except relD[row, relD_map[(row % Nq) * (DA - 96) + j]] =
   * bf16(dqa[row, 96 + j] * relD_scale) where the map is >= 0 -- what svit_relpos_scatter builds in a launch
   * of its own (bit-identical).  relD_ld % 8 == 0, relD_ld <= 544, relD 16-byte aligned. */
  void* relD; int32_t relD_ld; const int32_t* relD_map; float relD_scale;
  /* and, when relD_ld <= 128 (one 32-row pass), with relR bf16 [96, relD_ld] (the transposed concatenated tables)
   * the rel-pos backward's "dq = D R" GEMM is multiplied by the same kernel from the rows it builds:
   *   relX != NULL: dq_extra f32 [B*h*Nq, 96] = relD . relR^T is written there;
   *   relX == NULL ("fold"): the product is ADDED to dqa[:, 0:96] before it is rounded to bf16 -- dqa then is
   *   the whole gradient of the pooled q and svit_pool_ln_bwd needs no d_extra.
   * relR == NULL = off (the caller runs svit_gemm_nt on relD). */
  const void* relR; float* relX;
} svit_attn_bwd_args;
int svit_attn_bwd(const svit_attn_bwd_args* a, void* stream);
/* number of dk / dv planes svit_attn_bwd will write for these arguments (>= 1; only the shape fields,
 * DA and q_splits are read; q_splits > 0 asks for that many, the result is what is actually used);
 * negative = error code */
int svit_attn_bwd_parts(const svit_attn_bwd_args* a);

/* ---------------------------------------------------------- max-pool skip path (K7) ---- */
/* MaxPool3d((1,3,3),(1,2,2),(0,1,1)) on patch tokens, cls/objects copied
 * (attention.py:549-555,562-564). x f32 [B,N,C] -> y f32 [B,Nout,C]; idx uint8 (tap 0..8). */
int svit_maxpool_fwd(const float* x, float* y, uint8_t* idx, int B, int T, int H, int W,
                     int n_obj, int C, void* stream);
int svit_maxpool_bwd(const float* dy, const uint8_t* idx, float* dx, int B, int T, int H, int W,
                     int n_obj, int C, void* stream);
/* the same with dx rounded to bf16 [B,N,C]: what the dim-change projection's backward consumes
 * (attention.py:520-523 under autocast), without the f32 round trip */
int svit_maxpool_bwd_bf16(const float* dy, const uint8_t* idx, void* dx_bf16, int B, int T, int H, int W,
                          int n_obj, int C, void* stream);

/* ------------------------------------------------------------- optimiser tail (K17) ---- */
/* clip_grad_norm_(1.0) + AdamW (tools/train_net.py:144-151, models/optimizer.py:102-108)
 * on flat f32 buffers.  sumsq is a device scalar (pre-zeroed). */
/* sumsq += sum(g^2), deterministic (two-stage through `workspace`, >= 1024 floats): replicas
 * of a data-parallel job must derive the bit-identical clip coefficient. */
int svit_sumsq(const float* g, int64_t n, float* sumsq, float* workspace,
               int64_t workspace_floats, void* stream);
int svit_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq,
                    float max_norm, float lr, float beta1, float beta2, float eps, float wd,
                    int step, float grad_scale, void* stream);

/* ------------------------------------------------------------------ head (K15) ---------- */
/* SViT head (slowfast/models/video_model_builder.py:408-551) in one launch each way, fp32: dropout factors
 * applied to the cls / object rows of the final norm's output, class logits from the cls row, box MLP +
 * sigmoid and objectness logit from every object row, contact-state logits from the first two objects of a
 * frame.  tokens f32 [B,N,C]: row 0 = cls, the last T*O rows = objects (t-major).  keep f32 [B,1+T*O,C]
 * (mask / (1-p)) or NULL.  Outputs: logits [B,n_cls]; boxes [B,T,O,5] (col 0 objectness logit, 1..4 sigmoid
 * boxes = the reference's pred_bboxes); contact [B,T,2,5]; xobj [B,T*O,C] (optional: the dropped object
 * features = the reference's obj_desc). */
typedef struct {
  const float* tokens; int32_t B, N, C, T, O;
  const float* keep;
  const float* w_proj; const float* b_proj; int32_t n_cls;
  const float* w_box; const float* b_box;
  const float* w_bce; const float* b_bce;
  const float* w_con; const float* b_con;
  float* logits; float* boxes; float* contact; float* xobj;
} svit_head_args;
int svit_head_fwd(const svit_head_args* a, void* stream);
/* Backward: any of dlogits / dboxes / dcontact / dxobj may be NULL (no gradient).  dtokens f32 [B,N,C] is
 * OVERWRITTEN (zero rows for the patch tokens); the eight parameter gradients are ACCUMULATED (+=) -- they are
 * meant to be the views of the flat gradient buffer. */
typedef struct {
  svit_head_args f;
  const float* dlogits; const float* dboxes; const float* dcontact; const float* dxobj;
  float* dtokens;
  float* gw_proj; float* gb_proj; float* gw_box; float* gb_box; float* gw_bce; float* gb_bce; float* gw_con; float* gb_con;
} svit_head_bwd_args;
int svit_head_bwd(const svit_head_bwd_args* g, void* stream);

/* ------------------------------------------------ image-rank HAOG losses ---- */
/* boxes_loss_ + contact-state CE of VideoImageLoss._haog_loss (slowfast/models/losses.py:50-93,
 * 138-155; slowfast/utils/box_ops.py:10-77) without the reference's boolean-index host syncs:
 * masks stay arithmetic, shapes fixed, so the image rank's step replays as a HIP graph.
 * pred f32 [R,5] (objectness logit, sigmoid cx,cy,w,h), tar f32 [R,4] (all-zero row = empty),
 * contact f32 [Rc,5] logits, contact_tar int64 [Rc] (<0 ignored).  One launch writes
 * losses f32 [8] = {l1, bce, giou, contact CE, #boxes, #contacts, #targets>4, 0} and the unit
 * gradients g_l1/g_giou f32 [R,4], g_bce f32 [R], g_contact f32 [Rc,5]. */
int svit_haog_loss(const float* pred, const float* tar, const float* contact,
                   const int64_t* contact_tar, float* losses, float* g_l1, float* g_bce,
                   float* g_giou, float* g_contact, int R, int Rc, void* stream);
/* dpred [R,5] = up[1]*g_bce | up[0]*g_l1 + up[2]*g_giou ; dcontact = up[3]*g_contact, with
 * `upstream` f32 [4] on the device (d total / d {l1, bce, giou, contact}). */
int svit_haog_loss_bwd(const float* upstream, const float* g_l1, const float* g_bce,
                       const float* g_giou, const float* g_contact, float* dpred,
                       float* dcontact, int R, int Rc, void* stream);

/* Video-rank classification loss (round 6): nn.CrossEntropyLoss(reduction="mean") of VideoImageLoss
 * (slowfast/models/losses.py:121,158) and its unit gradient in one launch -- replaces log_softmax + nll_loss and their
 * backward kernels.  logits f32 [B,C], labels int64 [B] (-100 = ignored row, as torch's default ignore_index; any other
 * label outside [0,C) makes the loss NaN); loss f32 [1] = mean over the counted rows, dlogits f32 [B,C] = d loss / d logits. */
int svit_ce_loss(const float* logits, const int64_t* labels, int B, int C, float* loss, float* dlogits, void* stream);
/* The step's random draws in one launch (round 6; replaces torch.rand + add + floor + div of DropPath,
 * slowfast/models/common.py:46-59, and the head's nn.Dropout mask): scales f32 [n_blocks, per_block] =
 * floor(keep[b] + U) / keep[b], drop f32 [n_drop] in {0, 1/(1-p_drop)}.  Philox-4x32-10 keyed by state[0], draw number
 * state[1] advanced by the launch itself (state: uint64 [3] in device memory = {seed, draw, 0}), so a replayed HIP graph
 * draws fresh numbers every replay. */
int svit_step_draws(uint64_t* state, const float* keep, int n_blocks, int per_block, float* scales,
                    int n_drop, float p_drop, float* drop, void* stream);

/* ---------------------------------------- multi-view test ensemble ---------- */
/* TestMeter.update_stats (slowfast/utils/meters.py:303-336) for one batch, on the device: clip n
 * belongs to video clip_ids[n] / num_clips; video_preds[v] (f32 [V,C]) += preds[n] (mode 0, "sum")
 * or = max(.,.) (mode 1, "max") in BATCH ORDER (bit-identical to the reference's sequential
 * loop), video_labels[v] = labels[n], clip_count[v] += 1.  `repeat` folds the batch's clips of a
 * video cyclically that many times (de-duplicated NUM_ENSEMBLE_VIEWS).  err int32 [4]:
 * {clip ids out of range, label conflicts (the reference's assert), labels out of range, 0},
 * incremented, never cleared. */
int svit_ensemble_update(const float* preds, const int64_t* labels, const int64_t* clip_ids,
                         int N, int C, int num_clips, int num_videos, int mode, int repeat,
                         float* video_preds, int64_t* video_labels, int64_t* clip_count,
                         int* err, void* stream);
/* metrics.topks_correct (slowfast/utils/metrics.py:9-50): counts[i] += #videos whose label is
 * among the ks[i] best scores (ties: lower class index first).  ks int32 [nk] on the device,
 * nk <= 8; counts int32 [nk] pre-zeroed. */
int svit_topk_correct(const float* video_preds, const int64_t* video_labels, int V, int C,
                      const int* ks, int nk, int* counts, int* err, void* stream);
/* ------------------------------------------------ diagnostics (tools/ only) ------------------ */
/* Process-wide tuning knobs for the measurement scripts under tools/ and for the variant-parity tests (A/B runs of
 * kernel variants inside one process).  NOT part of the drop-in surface: the product path (svit_amd/engine.py,
 * bench.py) never calls them.  They are the ONLY way to change a heuristic: the library reads no environment
 * variable; all of them write one table (csrc/common.h SvitKnob, csrc/misc.hip), every selectable variant computes the
 * same results (none skips work), out-of-range values return SVIT_ERR_ARG, and svit_debug_reset() restores every
 * default -- tests and tools call it in their `finally`.  They replace nothing in the reference (it has no native
 * code). */
/* NT GEMM (csrc/gemm_nt.hip): key 0 = pipeline stages (2..4, 0 = heuristic), key 1 = forced tile/ring
 * configuration (-1 = heuristic, 0..10; 8 = the pre-ring v2 heuristic), key 2 = forced K-step (32 / 64, 0 = heuristic). */
int svit_debug_set(int key, int val);
/* grouped TN GEMM (csrc/gemm_tn.hip): cost-model constants of the row-chunk planner (x 0.01: microseconds per
 * k-step, TB/s of the fp32-atomic flush; <= 0 leaves a constant unchanged), and the tile mode (0: 128x96 only,
 * 1: per-problem heuristic fitted on isolated launches, 2: 128x192 everywhere (default), 3: 128x192 where
 * K % 192 == 0). */
int svit_debug_set_tn(int step_us_x100, int atomic_tbs_x100);
int svit_debug_set_tn_tile(int mode);
/* pooling (csrc/pool.hip): key 0 = forward path of the small planes: 0 streaming kernels, 1 VALU slab conv, 2 (default)
 * MFMA conv where it is ahead (blocks 4-13 of 16x224^2) and the slab elsewhere, 3 MFMA conv wherever its geometry holds;
 * key 1 = conv backward: 1 (default) the fused plane-walk kernel (conv dgrad + conv wgrad in one launch),
 * 0 refuse it (the two streaming launches in a -DSVIT_DIAG_POOL_STREAMING build, SVIT_ERR_SHAPE in the product library); key 2 = forward of the planes past 14x14 (blocks 0-3): 1 (default) the staged conv (input
 * staged once in LDS) + the row-wise LayerNorm launch, 0 the streaming kernel; key 3 = one-plane volumes (T = 1):
 * conv + LayerNorm in one launch from an LDS-staged plane -- 2 (default since round 6) in every T = 1 pass, also where
 * pre / mean / rstd are saved for a backward (image ranks), 1 in no-grad passes only (the frames pass), 0 never. */
int svit_debug_set_pool(int key, int val);
/* which path the last svit_pool_conv_bwd_qkv call took: 1 = the fused plane-walk kernel, 0 = not (planes that do not fit its LDS
 * plan, a workspace smaller than the plan's partial rows, key 1 = 0: SVIT_ERR_SHAPE in the product library, the two streaming
 * launches in a -DSVIT_DIAG_POOL_STREAMING build), -1 = no call yet. */
int svit_debug_pool_bwd_path(void);
/* attention: key 0 = dkv kernel form (0 heuristic, 1 four waves, 2 eight waves with query halves), key 3 = the
 * forward's T' = 1 tile for Nk <= 64 (1 on (default), 0 the generic kernel). */
int svit_attn_debug_set(int key, int val);
/* every knob above back to its default */
int svit_debug_reset(void);
#ifdef __cplusplus
}
#endif
#endif
