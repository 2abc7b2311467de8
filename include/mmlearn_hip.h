/*
 * mmlearn_hip.h -- C ABI of libmmlearn_hip.so (gfx950 / MI355X).
 *
 * The reference (VectorInstitute/mmlearn) is pure Python over ATen ops; it has no
 * FFI for this path, so every entry point below replaces a *sequence of ATen ops*
 * of the reference, cited per function (paths relative to the reference root).
 * The reference-side binding is the ctypes stub shown in INTEGRATION.md
 * (mmlearn_amd/_lib.py is that stub in this repo).
 *
 * Conventions
 *  - every pointer is caller-owned DEVICE memory (row-major, contiguous unless a
 *    leading dimension is given, 16-byte aligned); the library never allocates,
 *    frees or synchronises; all work is enqueued on `stream` (a hipStream_t
 *    passed as void*, NULL = default stream);
 *  - return value: 0 on success, negative on error; mmk_last_error() returns a
 *    thread-local message (the Python layer raises RuntimeError/ValueError);
 *  - re-entrant; the only global state is the optional profiling recorder.
 */
#ifndef MMLEARN_HIP_H
#define MMLEARN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMK_ABI_VERSION 8

/* element types of user tensors */
enum { MMK_F32 = 0, MMK_BF16 = 1, MMK_F16 = 2 };
/* arithmetic modes of the similarity GEMMs */
enum { MMK_COMPUTE_BF16 = 1, MMK_COMPUTE_F32 = 0 };

int mmk_abi_version(void);
const char* mmk_last_error(void);
/* device check: returns 0 when a gfx950 device is current, <0 (and an error text) otherwise */
int mmk_device_check(void);

/* ------------------------------------------------------------------ profiling
 * HIP-event recorder around every kernel launch (events are recorded on the
 * launch stream).  Used by bench.py for the roofline object.  Kernel ids: MMK_K_*.
 */
enum {
  MMK_K_MATCH = 0, MMK_K_PACK, MMK_K_TRANSPOSE, MMK_K_SIM_STATS, MMK_K_LSE_REDUCE, MMK_K_LOSS_COMBINE,
  MMK_K_SIM_GRAD, MMK_K_GRAD_GEMM, MMK_K_GRAD_FINALIZE, MMK_K_L2NORM, MMK_K_IJEPA_LOSS_FWD,
  MMK_K_IJEPA_LOSS_BWD, MMK_K_GATHER_ROWS, MMK_K_SCATTER_ROWS, MMK_K_PRED_ASSEMBLE, MMK_K_PRED_ASSEMBLE_BWD,
  MMK_K_EMA, MMK_K_MASK_INDEX, MMK_K_LAYERNORM_FWD, MMK_K_LAYERNORM_BWD, MMK_K_ACT, MMK_K_ATTN_FWD,
  MMK_K_ATTN_BWD, MMK_K_WGRAD, MMK_K_RECALL, MMK_K_CLIP_FUSED, MMK_K_MLP_GEMM, MMK_K_WIN_ATTN_FWD, MMK_K_WIN_ATTN_BWD, MMK_K_CLIP_BWD_FUSED, MMK_K_COUNT
};
int mmk_profile_enable(int on);          /* on=1 start recording (clears), on=0 stop */
/* resolve recorded events (synchronises on them); fills count[k] and total_ms[k] for k < MMK_K_COUNT */
int mmk_profile_read(int32_t* count, double* total_ms);
const char* mmk_kernel_name(int kernel_id);

/* ------------------------------------------------------------------ matching
 * Replaces find_matching_indices (mmlearn/datasets/core/example.py:101-166):
 * all (i, j) with ids_a[i] == ids_b[j] (both int64 columns), in row-major
 * (torch.where) order; duplicates give several pairs.
 *   workspace : int32[mmk_match_workspace_ints(n_a, n_b)]
 *   status    : int32[4]  -> {total matches R, 1 if the pairing is the identity
 *               (R == n_a == n_b and idx_a[p] == idx_b[p] == p), 1 if idx_a has
 *               repeated entries, 1 if idx_b has repeated entries}
 *   idx_a/idx_b : int32[capacity]; pairs beyond `capacity` are dropped (status[0]
 *               still holds the true total, so the caller can detect overflow).
 */
int mmk_match_workspace_ints(int n_a, int n_b);
int mmk_match_ids(const int64_t* ids_a, int n_a, const int64_t* ids_b, int n_b, int32_t* workspace,
                  int32_t* idx_a, int32_t* idx_b, int capacity, int32_t* status, void* stream);

/* ------------------------------------------------------------------ packing
 * Gather + (optional) L2 normalise + cast of embedding rows into the GEMM operand
 * layout.  Replaces F.normalize (mmlearn/modules/losses/contrastive.py:89-90) and
 * the advanced-index gathers emb[indices] (:290-291, :318-319).
 *   src  : [n_src, d] of src_dtype; idx: int32[r] or NULL (identity)
 *   dst  : [r_pad, k_pad] of compute type (bf16 or f32), rows >= r and columns >= d zero
 *   dstT : [k_pad, ldt] transpose of dst (NULL to skip), ldt >= r_pad, zero padded
 *   norm_out : float[r_pad] L2 norm of every packed row (of the values as rounded to the compute type; 0 for
 *          the padding rows), or NULL.  The forward uses them to bound the logits of a tile (mmk_clip_dir.x_norm).
 * k_pad is a multiple of 64 (bf16) / 32 (f32); r_pad a multiple of 128.
 */
int mmk_pack_rows(const void* src, int src_dtype, int n_src, int d, const int32_t* idx, int r, int normalize,
                  void* dst, void* dstT, int r_pad, int k_pad, int ldt, int compute, float* norm_out, void* stream);

/* ------------------------------------------------------------------ CLIP loss
 * One "direction" = CE over the rows of  s * X @ Y^T  with label(i) = label_off + i.
 * A LossPairSpec with local rows gives two directions (a->b, b->a), the reference's
 * logits_per_feature_a / logits_per_feature_b (contrastive.py:327-340).
 */
typedef struct {
  const void* x;      /* packed owned rows   [>= r rows, k_pad]            */
  const void* y;      /* packed all columns  [>= c rows, k_pad]            */
  const void* yT;     /* transpose of y      [k_pad, ldt] (backward only; may be NULL for a direction mmk_clip_backward_plan marks fused) */
  int32_t r;          /* owned rows                                         */
  int32_t c;          /* columns                                            */
  int32_t label_off;  /* positive column of row i is label_off + i         */
  int32_t ldt;        /* leading dimension of yT                            */
  float* part;        /* fwd workspace: float2[n_col_tiles * r] (ref,sum)   */
  float* diag;        /* fwd out: float[r] positive logit                    */
  float* lse;         /* fwd out: float[r] row log-sum-exp                   */
  float* loss_part;   /* fwd out: float[ceil(r/64)] sums of (lse_i - diag_i) over blocks of 64 rows */
  /* backward */
  const float* lse_col; /* float[c]: LSE of the opposite direction for every column */
  void* g;            /* bwd workspace: [r_pad, ldg] compute type (may be NULL for a fused direction) */
  int32_t ldg;        /* >= round_up(c, 128)                                 */
  float c_row, c_col, c_diag;     /* G = c_row*P_row + c_col*P_col - c_diag*[j == label] */
  float s_row, s_col, s_diag;     /* same triple for the d/dscale reduction               */
  float kappa;        /* w / (2 * rows in the mean) (times W for gather_with_grad), applied to dX */
  float ds_kappa;     /* same factor for the d/dscale reduction */
  float* slab;        /* bwd workspace: float[n_split, r_pad, k_pad]          */
  float* ds_part;     /* bwd workspace: float[n_grad_blocks]                  */
  /* finalize: scatter dX back to the user gradient */
  void* dx;           /* [n_src, d] user-dtype gradient buffer (pre-zeroed by caller) */
  const int32_t* dx_rows; /* int32[r] destination row per owned row, or NULL (identity) */
  int32_t dx_dtype;
  int32_t dx_accumulate; /* 1: destination rows may repeat -> atomic adds into f32 dx */
  const void* src;    /* original rows [n_src, d] (only when normalize=1, for the L2-norm backward) */
  int32_t src_dtype;
  int32_t normalize;
  /* mode 0: cross-entropy direction (above).  mode 1: modality-alignment BCE rows
   * (_compute_modality_alignment_loss, contrastive.py:344-413): y = all concatenated features [c rows], x = the
   * owned rows label_off .. label_off + r of y, hmax[row] = end of the positive columns [row, hmax[row]);
   * loss_part = block sums of (pos_r/npos_r + neg_r/nneg_r); backward writes the symmetrised d/dlogits. */
  int32_t mode;
  const int32_t* hmax;
  /* Mirrored direction.  On one rank the two directions of a pair see transposed logits (logits_per_b = logits_per_a^T,
   * contrastive.py:327-340): with mirror_part set (mode 0, r == c, label_off == 0) the forward takes the mirrored
   * direction's row statistics = this direction's COLUMN statistics from the same similarity tiles -- one GEMM pass per
   * pair instead of two -- and fills mirror_lse / mirror_loss_part like lse / loss_part of a direction with x and y
   * swapped.  Backward: with gT set the gradient-tile pass also stores G^T, which is the mirrored direction's G; that
   * direction is then passed with g = this gT and g_ready = 1 and skips its own tile pass. */
  float* mirror_part;      /* fwd workspace: float2[mmk_clip_mirror_tiles(r) * c] */
  float* mirror_lse;       /* fwd out: float[c] */
  float* mirror_loss_part; /* fwd out: float[ceil(c/64)] */
  void* gT;                /* bwd (optional): [round_up(c,128), ldgt] compute type */
  int32_t ldgt;            /* >= round_up(r, 128) */
  int32_t g_ready;         /* bwd: g already holds this direction's G (written as another direction's gT) */
  /* fwd (optional): L2 norms of the packed rows of x / y (mmk_pack_rows norm_out).  With both set, an interior tile whose
   * logits are bounded by |s| log2(e) max|x_i| max|y_j| <= 48 takes ONE exponential per element for the row and the column
   * statistics and searches no maximum; without them every tile takes per-row / per-column maxima. */
  const float* x_norm;
  const float* y_norm;
  /* bwd, alternative to gT for the second direction of a mirrored pair (bf16 compute): g = the FIRST direction's G
   * [c rows, ldg >= round_up(r, 128)] with g_ready = 1 and g_transposed = 1; dX = g^T Y is then formed by the
   * weight-gradient kernel (contraction over the rows of both operands, csrc/wgrad.hip) into tn_ws
   * (float[tn_ws_floats >= mmk_wgrad_plan(c, round_up(r, 128), k_pad)]); yT and slab are not used, G^T is never stored. */
  int32_t g_transposed;
  float* tn_ws;
  int64_t tn_ws_floats;
} mmk_clip_dir;

/* rows of the mirror_part workspace for r owned rows */
int mmk_clip_mirror_tiles(int r);

/* Several operands packed by ONE launch (gather + optional L2 normalise + cast to the compute type + zero pad, and the
 * transposed copy when dstT != NULL): the operands of a pair, which mmk_pack_rows would pack with two launches each. */
typedef struct mmk_pack_req {
  const void* src;      /* [n_src, d] rows of the user dtype */
  const int32_t* idx;   /* int32[r] source row per packed row, or NULL (identity) */
  void* dst;            /* [r_pad, k_pad] compute type */
  void* dstT;           /* [k_pad, ldt] compute type, or NULL */
  int32_t r, r_pad, normalize, ldt;
  float* norm;          /* float[r_pad] L2 norms of the packed rows, or NULL */
} mmk_pack_req;
int mmk_pack_rows_many(const mmk_pack_req* reqs, int n, int src_dtype, int d, int k_pad, int compute, void* stream);

/* tile configuration query: n_col_tiles for `part`, blocks for `ds_part`, split-K factor for `slab` */
int mmk_clip_plan(int r, int c, int k_pad, int compute, int32_t* n_col_tiles, int32_t* n_grad_blocks,
                  int32_t* n_split);

/* forward: similarity tiles + per-tile softmax statistics + positive logit, then the merge of the tile partials into the
 * row log-sum-exp (K5-K6 of SURVEY 2.3: _safe_matmul, logit_scale*, F.cross_entropy of contrastive.py:134-144,327-340): two
 * launches for all directions; scale is a device float (the reference's 0-dim logit_scale).
 * `tickets`: 4-byte slots [n_tickets >= mmk_clip_tickets(dirs, n_dirs)], only used by mmk_clip_forward_loss; slot 0 is a
 * counter that must be ZERO on entry and is zero again when the call's launches have completed (one buffer per stream can
 * be reused call after call), the other slots need no initialisation. */
int mmk_clip_tickets(const mmk_clip_dir* dirs, int n_dirs);
int mmk_clip_forward(const mmk_clip_dir* dirs, int n_dirs, int k_pad, int d, int compute, const float* scale,
                     int32_t* tickets, int n_tickets, void* stream);
/* The same, and the weighted loss value  sum_k loss_w[k] * sum_i (lse_i - diag_i)  (contrastive.py:134-144,160) is written
 * to loss_out (device float) by the merge launch (the workgroup that draws the last ticket adds the block sums in a fixed
 * order): loss_w (host) holds one weight per direction, followed by its mirror's when it has one. */
int mmk_clip_forward_loss(const mmk_clip_dir* dirs, int n_dirs, int k_pad, int d, int compute, const float* scale,
                          const float* loss_w, float* loss_out, int32_t* tickets, int n_tickets, void* stream);
/* separate=0: out[0] = sum_k weights[k] * sum(ptrs[k][0..counts[k]))  -- the loss value
 * (contrastive.py:134-144,160; weight = w / (2 * rows in the mean));  separate=1: out[k] = weights[k] * sum(ptrs[k]) */
int mmk_reduce_sums(const float* const* ptrs, const int32_t* counts, const float* weights, int n, int separate, float* out,
                    void* stream);
/* backward: recompute tiles -> G, dX = G @ Y (split-K), scale/normalise/scatter; dscale accumulated into
 * dscale_out (float[1], pre-zeroed by the caller); upstream is the device scalar dL/dloss. */
int mmk_clip_backward(const mmk_clip_dir* dirs, int n_dirs, int k_pad, int d, int compute, const float* scale,
                      const float* upstream, float* dscale_out, void* stream);
/* Which directions of such a call run as ONE kernel that recomputes its gradient tiles on chip (csrc/clip_bwd.hip: row-sharded
 * directions, R owned rows against C >> R gathered columns -- what every rank runs at W > 1): fused[k] = 1 means direction k
 * reads neither `yT` nor `g` (both may be NULL in the call) and leaves `g` unwritten.  Decided from the shapes, the coefficient
 * mode and the pairing of the descriptors only, so the caller can ask before it builds the transposed operand: no buffer is read,
 * x, y, yT and the workspaces may be unset.  A mirrored pair is described as in the real call -- the second direction has
 * g_ready = 1 and the SAME `g` value as the first -- but `g` is only compared here, any distinct non-null tag will do.
 * No device work. */
int mmk_clip_backward_plan(const mmk_clip_dir* dirs, int n_dirs, int k_pad, int compute, int32_t* fused);

/* ------------------------------------------------------------------ CLIP loss, one resident-grid launch (small batches)
 * The whole of ContrastiveLoss.forward for up to four LossPairSpecs on ONE rank -- emb[indices] (contrastive.py:290-291),
 * logit_scale * _safe_matmul (:339-340), both F.cross_entropy terms / 2 * weight (:134-144), torch.stack().sum() (:160) --
 * AND the gradients of that loss, in one launch (csrc/clip_fused.hip): similarity tiles in registers, tile statistics handed
 * over through L2, G = P_row + P_col - 2 delta from the same accumulators, dA = G B and dB = G^T A as raw f32 sums in the
 * workspace.  mmk_clip_fused_backward is one more launch: x weight / (2 n) * scale * upstream, cast, scatter through idx.
 * Conditions: 1..1024 matched rows per pair, f32 or bf16 rows of whole 16-byte pieces (d % 4 / d % 8 == 0), bf16 MFMA
 * arithmetic with f32 accumulation (what the reference computes under Lightning's bf16-mixed), and a grid that is
 * co-resident: `grid` <= `capacity` of mmk_clip_fused_plan (the workgroups of a launch wait for each other).
 *   a / b      : the embedding matrices [*, d] of the pair's two modalities (src_dtype), untouched
 *   idx_a/idx_b: int32[n] matched-row lists of find_matching_indices, or NULL (identity: pair p = rows (p, p))
 *   ws         : workspace of mmk_clip_fused_plan's ws_bytes; all ZERO before the first call, then reusable call after call
 *                on one stream (its counters are zero again when a launch has completed); it holds the raw gradient sums
 *                until mmk_clip_fused_backward has run
 *   loss_out   : float[1];  ds_out: float[2] = {raw d loss / d scale, 0} (NULL when want_grad == 0); the zero is a ready-made
 *                accumulator for mmk_clip_fused_backward's dscale_out (which ADDS upstream * ds_raw)
 * A spin that exceeds its bound (a workgroup of the launch never became resident) ends with NaN in loss_out / ds_out. */
typedef struct mmk_fused_pair {
  const void* a;
  const void* b;
  const int32_t* idx_a;
  const int32_t* idx_b;
  int32_t n;              /* matched pairs */
  float weight;           /* LossPairSpec.weight */
  /* backward only */
  void* da;               /* [rows of a, d] gradient buffer (dx_dtype; f32 and pre-zeroed when da_accumulate) */
  void* db;
  int32_t da_accumulate;  /* 1: idx_a repeats rows, or another pair writes the same buffer -> atomic adds */
  int32_t db_accumulate;
} mmk_fused_pair;
int mmk_clip_fused_plan(const int32_t* n, int n_pairs, int d, int src_dtype, int64_t* ws_bytes, int32_t* grid, int32_t* capacity);
int mmk_clip_fused_forward(const mmk_fused_pair* pairs, int n_pairs, int d, int src_dtype, const float* scale, void* ws, int64_t ws_bytes,
                           int want_grad, float* loss_out, float* ds_out, void* stream);
int mmk_clip_fused_backward(const mmk_fused_pair* pairs, int n_pairs, int d, int dx_dtype, const float* scale, const float* upstream,
                            void* ws, int64_t ws_bytes, const float* ds_raw, float* dscale_out, void* stream);
/* measurement hook: in a -DMMK_DEBUG_SWITCHES build, later launches write [grid][8] realtime-clock stamps (100 MHz) of every
 * workgroup's phases to device_buf (NULL switches it off); the product build accepts NULL only (tools/fused_phases.py) */
int mmk_clip_fused_debug_stamps(unsigned long long* device_buf);

/* ------------------------------------------------------------------ row ops
 * F.normalize(x, p=2, dim=-1, eps=1e-12) forward / backward
 * (mmlearn/tasks/contrastive_pretraining.py:428-429; modules/layers/normalization.py:34). */
int mmk_l2norm_fwd(const void* x, void* y, float* inv_norm, int rows, int d, int dtype, void* stream);
/* the same with a second output: y16 [rows, d] bf16 = y rounded (nullable).  The task's embeddings leave F.normalize as f32
 * (autocast keeps it there) and the bf16 similarity kernels round them again on every read; with the twin written by the
 * producer the one-launch loss (mmk_clip_fused_forward) reads half the bytes.  Same values bit for bit. */
int mmk_l2norm_fwd_twin(const void* x, void* y, void* y16, float* inv_norm, int rows, int d, int dtype, void* stream);
int mmk_l2norm_bwd(const void* x, const void* dy, const float* inv_norm, void* dx, int rows, int d, int dtype,
                   void* stream);

/* ------------------------------------------------------------------ I-JEPA
 * mask -> sorted keep indices (replaces the boolean-mask nonzero of apply_masks,
 * mmlearn/datasets/processors/masking.py:264-283).  mask: int32[b, n] of 0/1; idx: int32[b, keep];
 * bad: int32[1] set to 1 if any row's popcount != keep. */
int mmk_mask_to_index(const int32_t* mask, int b, int n, int keep, int32_t* idx, int32_t* bad, void* stream);

/* out[m*b + bi, p, :] = x[bi, idx[m, bi_or_0, p], :]   (apply_masks, masking.py:241-287)
 * idx: int32[n_masks, idx_b, keep] with idx_b == b (per-sample) or 1 (batch-shared). */
int mmk_gather_rows(const void* x, void* out, const int32_t* idx, int b, int n, int d, int n_masks, int idx_b,
                    int keep, int dtype, void* stream);
/* backward of gather: dx[bi, idx[...], :] += dout[...] ; dx pre-zeroed by the caller */
int mmk_scatter_rows(const void* dout, void* dx, const int32_t* idx, int b, int n, int d, int n_masks, int idx_b,
                     int keep, int dtype, void* stream);

/* fused target path + regression loss (tasks/ijepa.py:232-238,250-261):
 *   t = layer_norm(h[bi, idx[m,.,p], :])  (no affine, eps) ; loss = mean(rho(z - t))
 * kind 0: smooth-L1 (beta=1, the reference default), kind 1: MSE.
 * z: [n_masks*b, keep, d]; h: [b, n, d]; target_out (optional) receives t in z's dtype;
 * part: float[n_blocks] workspace; loss: float[1]. */
int mmk_ijepa_loss_fwd(const void* z, const void* h, const int32_t* idx, int b, int n, int d, int n_masks, int idx_b,
                       int keep, int dtype, int kind, float eps, void* target_out, float* part, int n_blocks,
                       float* loss, void* stream);
int mmk_ijepa_loss_blocks(int rows);
/* dz = upstream * rho'(z - t) / numel, t recomputed from h */
int mmk_ijepa_loss_bwd(const void* z, const void* h, const int32_t* idx, int b, int n, int d, int n_masks, int idx_b,
                       int keep, int dtype, int kind, float eps, const float* upstream, void* dz, void* stream);

/* predictor sequence assembly (mmlearn/modules/encoders/vision.py:545-560):
 *   seq[m*b*ne + r, :n_ctxt]  = x[r] + pos[enc_idx[e(r), bi(r), :]]
 *   seq[m*b*ne + r, n_ctxt:]  = mask_token + pos[pred_idx[m, bi, :]]
 * x: [ne*b, n_ctxt, d]; pos: [n, d]; mask_token: [d]; seq: [np*ne*b, n_ctxt + n_pred, d]. */
int mmk_pred_assemble(const void* x, const void* pos, const void* mask_token, const int32_t* enc_idx,
                      const int32_t* pred_idx, int b, int n, int d, int n_enc, int n_pred_masks, int enc_idx_b,
                      int pred_idx_b, int n_ctxt, int n_pred, int dtype, void* seq, void* stream);
/* dx[r] = sum_m dseq[m*b*ne + r, :n_ctxt];  dtok_part[blk, :] partial sums of dseq[:, n_ctxt:, :] (float) */
int mmk_pred_assemble_bwd(const void* dseq, int b, int d, int n_enc, int n_pred_masks, int n_ctxt, int n_pred,
                          int dtype, void* dx, float* dtok_part, int n_tok_blocks, void* dtok, void* stream);
int mmk_pred_tok_blocks(int rows);

/* multi-tensor EMA / copy (mmlearn/modules/ema.py:132-158).  table: device array of n_tensors
 * records {teacher ptr, student ptr, numel, teacher dtype, student dtype}; mode 0: teacher = student
 * (the reference's observable behaviour), mode 1: teacher = decay*teacher + (1-decay)*student
 * computed in f32. */
typedef struct {
  void* teacher;
  const void* student;
  int64_t numel;
  int32_t teacher_dtype;
  int32_t student_dtype;
} mmk_ema_entry;
int mmk_ema_update(const mmk_ema_entry* table, int n_tensors, int64_t max_numel, float decay, int mode,
                   void* stream);
/* The same launch with the decay read from a device word (decay_dev[0], f32): a step captured into a HIP graph replays with
 * the value the word holds at replay time (the annealed schedule of modules/ema.py:79-89,166-177 advances on the host and is
 * written into the word between replays).  ABI 5. */
int mmk_ema_update_dev(const mmk_ema_entry* table, int n_tensors, int64_t max_numel, const float* decay_dev, int mode,
                       void* stream);

/* Multi-tensor AdamW step for one parameter group (torch.optim.AdamW semantics: decoupled weight decay, bias
 * correction, eps added to sqrt(v)/sqrt(bc2); amsgrad / maximize off).  f32 parameters and moments; gradients f32 or
 * bf16 (grad_dtypes[k] = MMK_*), their pointers re-supplied every step.  `chunks` is the flat work list: one entry per
 * mmk_adamw_chunk_elems() elements of a tensor.  All four tables live in device memory. */
typedef struct {
  void* param;
  void* exp_avg;
  void* exp_avg_sq;
  int64_t numel;
} mmk_adamw_tensor;
typedef struct {
  int32_t tensor;
  int32_t pad;
  int64_t offset;
} mmk_adamw_chunk;
int mmk_adamw_chunk_elems(void);
int mmk_adamw_update(const mmk_adamw_tensor* tensors, const void* const* grads, const int32_t* grad_dtypes, const mmk_adamw_chunk* chunks,
                     int n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream);
/* the same update with the learning rate and the step count (including this step, as f32) read from device memory and the bias
 * corrections formed in the kernel: no host scalar of the step is baked into the launch, so it can be captured into a HIP graph
 * (the plumb point of mmlearn/cli/run.py:139 -- whole-step capture / compile) and replayed */
int mmk_adamw_update_dev(const mmk_adamw_tensor* tensors, const void* const* grads, const int32_t* grad_dtypes, const mmk_adamw_chunk* chunks,
                         int n_chunks, const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, const float* step_dev,
                         void* stream);

/* ------------------------------------------------------------------ eval-side retrieval metric
 * RetrievalRecallAtK._process_batch + _recall_at_k (mmlearn/modules/metrics/retrieval_recall.py:239-289): for every
 * query row of x (f32 [n, d], L2-normalised) the number of rows of y (f32 [m, d]) whose score x_i . y_j beats the
 * score of the query's positive pos[i]; ties go to the lower database index.  recall@k of the row = (rank < k).
 * tpos_ws: f32[n] workspace; rank: int32[n] out. */
int mmk_recall_ranks(const float* x, const float* y, const int64_t* pos, float* tpos_ws, int32_t* rank, int n, int m, int d,
                     void* stream);

/* ------------------------------------------------------------------ encoder-side row ops
 * torch.nn.LayerNorm inside the encoders the tasks drive (mmlearn/modules/encoders/{clip,text,vision}.py):
 * F.layer_norm forward / backward with f32 statistics.  x: [rows, d]; w, b: f32[d] or NULL; mean/rstd: f32[rows].
 * fwd dtype = x dtype | (y dtype << 4);  bwd dtype = x(=dx) dtype | (dy dtype << 4).
 * bwd workspaces: part f32[mmk_layernorm_part_blocks(rows), 2, d], part2 f32[256, 2, d]; dw/db f32[d] or both NULL. */
int mmk_layernorm_part_blocks(long rows);
int mmk_layernorm_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int64_t rows, int d,
                      float eps, int dtype, void* stream);
int mmk_layernorm_bwd(const void* x, const void* dy, const float* w, const float* mean, const float* rstd, void* dx, float* part,
                      float* part2, float* dw, float* db, int64_t rows, int d, int dtype, void* stream);
/* Residual add (+ hidden-state dropout) + LayerNorm in one pass:  s = r + dropout(x),  y = LN(s).  Replaces
 * "hidden = residual + sublayer_out" followed by the next nn.LayerNorm in the encoders' transformer blocks (HF
 * CLIPEncoderLayer: layer_norm2(residual + attn); BertSelfOutput / BertOutput: LayerNorm(dropout(dense(h)) + input)).
 * x: [rows, d] in dtype&15, r / s: f32, y in dtype>>4.  s may be NULL only if the caller never needs the sum (it is
 * what the backward re-normalises, so training callers pass it).  Dropout: the counter-based mask of (seed, row, col).
 * xbias (nullable, f32[d]): the bias of the Linear that produced x, when that Linear was run without it -- it is added
 * before the dropout, and its gradient (column sums of dx) comes out of the backward as dxbias instead of a separate
 * dY.sum(0) pass.  y_twin (nullable): a bf16 copy of y for the GEMM that consumes it in post-LN blocks (BERT), where y
 * is also the f32 residual stream; the gradient reaching the twin comes back through dy_twin and is summed in-kernel. */
int mmk_add_layernorm_fwd(const void* x, const float* xbias, const float* r, const float* w, const float* b, float* s, void* y,
                          void* y_twin, float* mean, float* rstd, int64_t rows, int d, float eps, int dtype, float dropout_p,
                          uint64_t seed, void* stream);
/* Backward: ds = ds_in + LNbwd(dy) (ds_in nullable);  dr = ds (f32);  dx = dropout_mask(ds) in dtype&15;  dy in dtype>>4;
 * dw/db (nullable pair) with the workspaces of mmk_layernorm_bwd (3 slabs per block instead of 2 when dxbias is given). */
int mmk_add_layernorm_bwd(const float* s, const void* dy, const void* dy_twin, const float* ds_in, const float* w, const float* mean,
                          const float* rstd, float* dr, void* dx, float* part, float* part2, float* dw, float* db, float* dxbias,
                          int64_t rows, int d, int dtype, float dropout_p, uint64_t seed, void* stream);

/* y = act(x + bias) for a Linear run without its bias (fc1 of the encoders' MLPs: HF CLIPMLP fc1 + quick_gelu,
 * BertIntermediate dense + erf GELU); act 0 = x*sigmoid(1.702x), 1 = erf GELU.  The backward writes
 * dx = act'(x + bias) * dy and dbias = column sums of dx (part: f32[mmk_bias_act_part_blocks(rows), d], part2: f32[256, d]). */
int mmk_bias_act_part_blocks(long rows);
int mmk_bias_act_fwd(const void* x, const float* bias, void* y, int64_t rows, int d, int act, int dtype, void* stream);
int mmk_bias_act_bwd(const void* x, const float* bias, const void* dy, void* dx, float* part, float* part2, float* dbias, int64_t rows,
                     int d, int act, int dtype, void* stream);
/* column sums of a f32 [n_rows, d] buffer of partial rows (fixed summation order): out f32[d]; part2: f32[256, d] scratch */
int mmk_colsum_f32(const float* part, int n_rows, int d, float* part2, float* out, void* stream);
/* out[n] (f32) = column sums of x [rows, n] (bf16, n % 8 == 0, or f32, n % 4 == 0; contiguous): `grad_output.sum(0)`, the bias gradient
 * of every nn.Linear whose bias no neighbouring kernel takes care of (HTSAT's, under mmlearn/modules/encoders/).  part: f32 workspace
 * [mmk_colsum_rows_slices(rows), n]; fixed summation order. */
/* F.interpolate(x, (h_out, w), mode="bicubic", align_corners=True) for an input whose last axis keeps its length -- the spectrogram
 * stretch of HF ClapAudioEncoder.reshape_mel2img (HTSAT, BASELINE configs[3]): x f32 [n_img, h_in, w] -> y [n_img, h_out, w], ATen's
 * taps and arithmetic along the one axis that changes; backward != 0 computes the input gradient from the output gradient. */
int mmk_cubic_resize_rows(const float* x, float* y, int64_t n_img, int h_in, int h_out, int w, int backward, void* stream);
int mmk_colsum_rows_slices(int64_t rows);
int mmk_colsum_rows(const void* x, int64_t rows, int n, int dtype, float* part, float* out, void* stream);

/* The two GEMMs of an encoder MLP that sit next to its activation, with the activation pass in the epilogue (csrc/mlp_gemm.hip).
 * Replaces, in the reference's op sequence (mmlearn/modules/layers/mlp.py; HF CLIPMLP / BertIntermediate+BertOutput under
 * mmlearn/modules/encoders/clip.py:29-470, text.py:20-178, and their autograd):
 *   forward   F.linear(x, W1) -> + b1 -> activation                       H = act(X W1^T + b1), pre = X W1^T (bf16, bias-free)
 *   backward  grad_out @ W2 -> * act'(pre + b1) -> .sum(0) for b1.grad    dPre = (dY Wt^T) * act'(pre + b1), part = column sums
 * All matrices bf16 with K-contiguous rows (strides in elements, multiples of 8); f32 accumulation; act 0 = x*sigmoid(1.702x),
 * 1 = erf GELU (numbering of mmk_bias_act_*).  Shapes: M % 256 == 0, N % 256 == 0, K % 64 == 0 (mmk_mlp_gemm_supported);
 * other shapes stay on library GEMM + mmk_bias_act_*.  Wt = fc2.weight^T as [N = hidden, K = out] (the dX twin of
 * mmk_cast_transpose).  part (nullable): f32[mmk_mlp_gemm_part_rows(M)][N], reduce with mmk_colsum_f32 -> d b1. */
int mmk_mlp_gemm_supported(int64_t M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc);
int mmk_mlp_gemm_part_rows(int64_t M);
int mmk_mlp_gemm_plain(const void* A, const void* B, void* C, int64_t M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, void* stream);
int mmk_mlp_gemm_fwd_act(const void* X, const void* W, const float* bias, void* H, void* pre, int64_t M, int N, int K, int64_t ldx,
                         int64_t ldw, int64_t ldc, int act, void* stream);
int mmk_mlp_gemm_bwd_dact(const void* dY, const void* Wt, const void* pre, const float* bias, void* dPre, float* part, int64_t M, int N,
                          int K, int64_t ldy, int64_t ldw, int64_t ldp, int64_t ldc, int act, void* stream);
/* The pair the product runs: the forward leaves G = act'(X W^T + bias) behind instead of the pre-activation (same bytes; three more
 * instructions per element next to the ones act needs), the backward multiplies by it: dPre = (dY Wt^T) * G, part = column sums. */
int mmk_mlp_gemm_fwd_act_grad(const void* X, const void* W, const float* bias, void* H, void* G, int64_t M, int N, int K, int64_t ldx,
                              int64_t ldw, int64_t ldc, int act, void* stream);
int mmk_mlp_gemm_bwd_mul(const void* dY, const void* Wt, const void* G, void* dPre, float* part, int64_t M, int N, int K, int64_t ldy,
                         int64_t ldw, int64_t ldg, int64_t ldc, void* stream);


/* Weight gradient of a Linear, dW[N, K] = dY^T x for dY [M, N], x [M, K] (bf16, row strides ldy / ldx in elements):
 * replaces the dY.t() @ x GEMM of autograd's linear backward where its output is too small to fill the chip
 * (attention output projections: 768 x 768 over M = 201,728).  Split over M, partial tiles in `ws`
 * (mmk_wgrad_plan gives the element count), summed into dw (out_dtype, row stride ldw). */
int mmk_wgrad_plan(int64_t M, int N, int K, int* splits_out, int64_t* ws_floats_out);
/* debugging: segment clocks of the 8-phase weight-gradient kernel (workgroup 0: 8 waves x {MFMA cluster, wait at the closing barrier, load
 * segment, wait at the opening barrier, loop clocks, loop time in 100 MHz ticks, K-tiles, -}); needs a library built with
 * -DMMK_WGRAD_STAMPS_BUILD and MMK_WGRAD_STAMPS=1 in the environment (tools/wgrad_stamps.py).  Synchronises the device. */
int mmk_wgrad_debug_stamps(unsigned long long* out);

/* the split partial tiles only: ws[split][n_pad][k_pad] f32 (n_pad / k_pad = N / K rounded up to 256); the caller sums */
int mmk_wgrad_partial(const void* dy, const void* x, float* ws, int64_t M, int N, int K, int64_t ldy, int64_t ldx,
                      int32_t* splits_out, int32_t* n_pad_out, int32_t* k_pad_out, void* stream);
int mmk_wgrad(const void* dy, const void* x, void* dw, float* ws, int64_t M, int N, int K, int64_t ldy, int64_t ldx, int64_t ldw,
              int out_dtype, void* stream);

/* im2col of a Conv2d whose stride equals its kernel (ViT patch embedding; HF CLIPVisionEmbeddings.patch_embedding,
 * mmlearn/modules/layers/embedding.py PatchEmbed): out[(b, py, px)][(c, i, j)] = in[b][c][py P + i][px P + j] as bf16,
 * so that the convolution runs as one GEMM against weight.view(E, C P P).  in: [B, C, H, W] of `dtype`, contiguous. */
int mmk_patchify(const void* in, void* out, int B, int C, int H, int W, int P, int dtype, void* stream);
/* its backward with respect to the image (the inverse permutation; a non-overlapping convolution's col2im has no sums):
 * din[b][c][py P + i][px P + j] = dcols[(b, py, px)][(c, i, j)].  dtype = dcols dtype (bf16 / f32) | image dtype << 4. */
int mmk_unpatchify(const void* dcols, void* din, int B, int C, int H, int W, int P, int dtype, void* stream);

/* bf16 operands of an encoder nn.Linear from its master weight w [n, k] of `dtype` (f32 / bf16 / f16), in one pass: w16 [n, k]
 * (may be null) for the forward x w16^T -- what autocast's per-step cast of the weight produces -- and w16t [k, n] = w16^T, so that
 * the backward's dX = dY w (the `grad_output @ weight` of every nn.Linear under mmlearn/modules/encoders/clip.py:29-470,
 * text.py:20-178) can run as a "x W^T"-layout product on dY and w16t.  n, k multiples of 4. */
int mmk_cast_transpose(const void* w, void* w16, void* w16t, int n, int k, int dtype, void* stream);

/* Backward of nn.Embedding (HF BertEmbeddings word / token-type tables): dw[ids[r], :] += dout[r, :], f32 dw zeroed by
 * the caller; runs of equal ids are summed in registers before one hardware float atomic per element (summation order is
 * not fixed: results can differ in the last bits between runs, as with ATen's atomic paths). */
int mmk_embedding_bwd(const void* dout, const int64_t* ids, float* dw, int64_t rows, int d, int64_t vocab, int dtype, void* stream);
/* The same gradient for many rows (mmlearn/modules/encoders/text.py:20-178: BERT's word and token-type tables at 78,848 token rows
 * per step) on ids sorted by the caller: ids_sorted non-decreasing, perm[r] = the row of dout that sorted position r came from.
 * A wave sums the runs of 32 consecutive sorted rows; runs strictly inside a chunk are plain stores, the chunk's first / last run go
 * into an (id, partial row) list that is sorted again and 16 times shorter, on which the kernel recurses; only the last, short
 * list (<= 1024 entries) ends in float atomics.  No same-address atomic storms for hot ids (token-type ids are all equal).
 * scratch: mmk_embedding_bwd_scratch_bytes(rows, d) bytes; dw f32 [vocab, d] zeroed by the caller; ids outside [0, vocab) ignored. */
int64_t mmk_embedding_bwd_scratch_bytes(int64_t rows, int d);
int mmk_embedding_bwd_sorted(const void* dout, const int64_t* ids_sorted, const int64_t* perm, float* dw, void* scratch, int64_t rows,
                             int d, int64_t vocab, int dtype, void* stream);

/* Windowed self-attention of HTSAT / Swin blocks (HF ClapAudioSelfAttention.forward, the audio tower of BASELINE configs[3]; in the
 * reference it arrives through mmlearn/modules/encoders/ HF wrappers): per (window, head)  O = softmax(scale Q K^T + table) V  with
 * 64-token windows and head dim 24 or 32.  q / k / v / o / dout / dq / dk / dv: bf16 [B * nW, 64, H * dh] contiguous (window bw = b * nW + w; the backward needs neither o nor P:
 * delta = rowsum(P * dP));
 * table: f32 [nWt, H, 64, 64] = relative-position bias per head (nWt = 1) or bias + shifted-window mask per (window position, head)
 * (nWt = nW); lse2: f32 [B * nW, H, 64] (base-2 log of the softmax denominators, written by fwd, read by bwd); dtab_part: f32
 * [mmk_win_attn_blocks(B, nW, H), 64, 64], one partial sum of dS per workgroup -- block id -> head: (id >> 3) % H when
 * mmk_win_attn_blocks / H is a multiple of 8, id % H otherwise.  img_w > 0 (token-map mode): the eight tensors are [B, img_h * img_w, C] token
 * maps instead, window w = (wy, wx) of the map rolled by -shift is gathered on the way in and scattered back on the way out: the
 * torch.roll / window_partition / window_reverse / roll of ClapAudioLayer.forward as address arithmetic (img_h, img_w multiples of 8).
 * ld: row stride in elements of q / k / v and dq / dk / dv -- C, or 3 C when they are the three thirds of one packed [.., 3 C] projection
 * output / gradient buffer (the three Linears run as one GEMM each way); o and dout always have row stride C. */
int mmk_win_attn_supported(int tokens, int dh, int c);
int mmk_win_attn_blocks(int B, int nW, int H);
int mmk_win_attn_fwd(const void* q, const void* k, const void* v, const float* table, void* o, float* lse2, int B, int nW, int nWt, int H, int dh,
                     float scale, int img_h, int img_w, int shift, int ld, void* stream);
int mmk_win_attn_bwd(const void* q, const void* k, const void* v, const void* dout, const float* lse2, const float* table, void* dq, void* dk,
                     void* dv, float* dtab_part, int B, int nW, int nWt, int H, int dh, float scale, int img_h, int img_w, int shift, int ld,
                     void* stream);

/* HF QuickGELUActivation  x * sigmoid(1.702 x)  (CLIP MLP), forward and backward, n elements (multiple of 4) */
int mmk_quick_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream);
int mmk_quick_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream);

/* Key-padding mask of a batch -> the per-sample key bias records the attention kernels read: rec is f32 [B][256], entry (b, j) is added
 * to the base-2 logit of key j of sample b for every head and every query: 0 = attended, -1e30 = masked (finite: a sample whose keys are
 * all masked averages V uniformly, HF's additive finfo.min convention, instead of NaN), -inf for j >= L.  The text towers of the
 * reference always forward the tokenizer's mask (mmlearn/modules/encoders/text.py:160-165, clip.py:104-107, 329-346); HF turns it into
 * a [B, 1, L, L] / [B, 1, 1, L] tensor per call.  `mask` by `kind`:
 *   MMK_KEYMASK_LENGTHS   int32 [B]: keys 0 .. len-1 attended (right padding)
 *   MMK_KEYMASK_U8 / I32 / I64 / F32_KEEP   [B, L] rows mask_sb elements apart: nonzero = attended (torch.bool is U8)
 *   MMK_KEYMASK_F32_ADD / BF16_ADD          [B, L] additive mask in natural-log units (0 / finfo.min or any bias; clamped to >= -1e30)
 * One record serves every layer and both passes of a tower's step.  ABI 7. */
#define MMK_KEYMASK_LENGTHS 0
#define MMK_KEYMASK_U8 1
#define MMK_KEYMASK_I32 2
#define MMK_KEYMASK_I64 3
#define MMK_KEYMASK_F32_KEEP 4
#define MMK_KEYMASK_F32_ADD 5
#define MMK_KEYMASK_BF16_ADD 6
int mmk_attn_key_bias(const void* mask, int kind, int B, int L, int64_t mask_sb, float* rec, void* stream);

/* Short-sequence self-attention of the encoders' blocks (mmlearn/modules/layers/attention.py:60-75 materialises
 * softmax(QK^T); HF encoders call SDPA): out = softmax(scale * Q K^T + mask) V per (batch, head), bf16, head_dim 64, L <= 256.
 * q/k/v are [B, H, L, 64] views given by element strides {batch, head, row} (last dim contiguous);
 * out is [B, L, H, 64] contiguous; lse (f32 [B, H, L], natural log of the scaled scores' sum) feeds the backward.
 * dropout_p > 0 applies attention-probability dropout (BERT's attention_probs_dropout_prob) with a counter-based
 * keep mask that is a pure function of (seed, batch, head, query, key): the backward regenerates it from the same
 * seed, nothing is stored.
 * key_bias (may be NULL: no mask): the records of mmk_attn_key_bias for this batch -- a key-padding mask, F.scaled_dot_product_attention's
 * attn_mask [B, 1, 1, L].  causal != 0 (needs key_bias; an all-attended record for no padding) also masks key j > query i (HF CLIP's
 * text tower, clip.py:329-346). */
int mmk_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int B, int H, int L, int dh,
                 const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, float scale,
                 float dropout_p, uint64_t seed, const float* key_bias, int causal, void* stream);

/* Backward of mmk_attn_fwd: out / dout are [B, L, H, 64] contiguous, lse is the forward's [B, H, L]; dq / dk / dv are
 * [B, L, H, 64] views with element strides grad_strides = {batch, row} (heads 64 apart), so the three gradients can be
 * written straight into one packed [B, L, 3, H, 64] buffer for a fused QKV projection;
 * delta_ws an f32 workspace of B * H * 512 elements (per (batch, head): 256 scaled-LSE values and 256 rowsum(dout . out)).
 * colsum_part (optional, may be NULL; packed layout only): f32 [B * ceil(L / 32)][3][H][64] -- per 32-row tile, the column
 * sums of the dq / dk / dv values as stored (rounded to bf16); summed over its first dimension it is the bias gradient
 * `dY.sum(0)` of the fused QKV projection (reference: autograd of the q/k/v nn.Linear biases), without re-reading dY.
 * Only where mmk_attn_bwd_has_colsum(L) == 1 (all L <= 224 except 97..128); passing it elsewhere is an error.
 * key_bias / causal: the forward's.
 * Replaces autograd through the same reference expressions (softmax(QK^T)V backward). */
/* debugging: shader-clock stamps of workgroup 0 of the five-product backward (needs a library built with
 * -DMMK_ATTN_STAMPS_BUILD and MMK_ATTN_STAMPS=1 in the environment before the first backward call): 16 per item -- item start, loads issued, loads landed, after each of the NT + 1 step
 * barriers, after the final stores were issued -- for the workgroup's first 32 items.  Synchronises the device. */
int mmk_attn_debug_stamps(unsigned long long* out, int n);
int mmk_attn_bwd_has_colsum(int L);
int mmk_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                 float* delta_ws, void* dq, void* dk, void* dv, int B, int H, int L, int dh, const int64_t* q_strides,
                 const int64_t* k_strides, const int64_t* v_strides, const int64_t* grad_strides, float scale,
                 float dropout_p, uint64_t seed, float* colsum_part, const float* key_bias, int causal, void* stream);

/* Attention of ONE query per (sample, head) against L <= 256 keys (head dim 64, bf16): the token-0 row of the last layer of a tower
 * pooled at token 0 (mmlearn/modules/encoders/clip.py:463-470 reads last_hidden_state[:, 0, :]; HF CLIPEncoderLayer / BertLayer are the
 * callers' stock forms).  q, o, dout, dq: [B, H, 64] contiguous.  k, v: element (b, l, h, d) at base + b * kv_sb + l * kv_sl + h * 64 + d
 * (the halves of one packed [B, L, 2, H, 64] projection output: kv_sl = 2 H 64); dk, dv likewise with g_sb, g_sl.  lse2: f32 [B, H], the
 * base-2 log-sum-exp of the scaled logits (forward output, backward input).  dropout_p drops attention probabilities with the counter
 * -based mask of mmk_attn_fwd (query index 0), regenerated by the backward from the same seed.  Replaces
 * F.scaled_dot_product_attention(q[:, :, :1], k, v, attn_mask, dropout_p, scale) and its autograd.  key_bias (may be NULL): the
 * key-padding records of mmk_attn_key_bias.  ABI 7. */
int mmk_cls_attn_supported(int L, int dh);
int mmk_cls_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse2, int B, int H, int L, int dh, int64_t kv_sb,
                     int64_t kv_sl, float scale, float dropout_p, uint64_t seed, const float* key_bias, void* stream);
int mmk_cls_attn_bwd(const void* q, const void* k, const void* v, const void* dout, const float* lse2, void* dq, void* dk, void* dv, int B,
                     int H, int L, int dh, int64_t kv_sb, int64_t kv_sl, int64_t g_sb, int64_t g_sl, float scale, float dropout_p,
                     uint64_t seed, const float* key_bias, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMLEARN_HIP_H */
