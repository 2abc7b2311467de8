// Pieces shared by the contrastive-loss translation units (clip.hip: tiled multi-launch path; clip_fused.hip: the one-launch
// resident-grid path for small batches).
#pragma once
#include "common.h"

namespace mmk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));

constexpr int MAX_PROBS = 8;

// device-coherent accesses for data that one workgroup writes and another reads within the same launch (block sums of
// the merge kernel): sc1 stores / loads that do not linger in a non-coherent XCD L2
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float2* p, float2 v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_agent(const float2* p) {
  return __builtin_bit_cast(float2, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// raw v_exp_f32 (2^x); arguments here are <= 0 up to rounding, tiny results may flush -- harmless in a softmax sum
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Transpose-reduce over the 32 lanes of a half-wave: every lane holds CNT values (one per "column"), on return lane L holds
// in v[0] the reduction over the 32 lanes (same lane >> 5) of column L & (CNT - 1)  (CNT = 32: 31 shuffles instead of 160;
// CNT = 16: the xor-16 partner pairs are combined at the end, both lanes of a pair hold the same column).
template <int CNT, bool IS_MAX>
__device__ __forceinline__ float half_wave_transpose_reduce(float (&v)[CNT], int lane) {
  static_assert(CNT == 32 || CNT == 16, "16 or 32 columns");
  constexpr int STEPS = CNT == 32 ? 5 : 4;
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const bool bit = (lane >> s) & 1;
#pragma unroll
    for (int i = 0; i < (CNT >> (s + 1)); ++i) {
      const float keep = bit ? v[2 * i + 1] : v[2 * i];
      const float send = bit ? v[2 * i] : v[2 * i + 1];
      const float recv = __shfl_xor(send, 1 << s);
      v[i] = IS_MAX ? fmaxf(keep, recv) : keep + recv;
    }
  }
  if (CNT == 16) {
    const float o = __shfl_xor(v[0], 16);
    v[0] = IS_MAX ? fmaxf(v[0], o) : v[0] + o;
  }
  return v[0];
}

// ------------------------------------------------------------------ finalize (clip.hip)
// one wave per owned row: dy = coef * sum_splits slab ; optional F.normalize backward ; scatter to user grad
struct FinProb {
  const float* slab;
  long split_stride;
  int slab_ld;
  int r;
  float kappa;
  void* dx;
  const int32_t* dx_rows;
  int accumulate;
  const void* src;  // original rows (normalize backward)
  int normalize;
  int n_split;      // slabs to sum
  // further sources with coefficients of their own, summed into the same destination rows (the one-launch loss: two pairs that
  // share a modality, identity pairing): extra[k] [r][slab_ld], kappa_extra[k]
  const float* extra[3] = {nullptr, nullptr, nullptr};
  float kappa_extra[3] = {0.f, 0.f, 0.f};
  int n_extra = 0;
  int exclusive = 0;  // with accumulate: every destination element has ONE writer in this launch -> load + add + store, no atomics
};
struct FinBatch {
  FinProb p[MAX_PROBS];
  int d;
};
struct DsBatch {
  const float* part[MAX_PROBS];
  int n[MAX_PROBS];
  float kappa[MAX_PROBS];
  int n_probs;
};
// one launch: a grid row per direction, plus one (when dscale_out != null) for the d/dscale reduction; U = dx_dtype
int launch_grad_finalize(const FinBatch& fb, int n_dirs, int max_r, int ld_max, const float* scale, const float* upstream, const DsBatch& db,
                         float* dscale_out, int dx_dtype, hipStream_t st);

// ------------------------------------------------------------------ one-kernel recompute-G backward of row-sharded directions (clip_bwd.hip)
struct BwdFusedProb {
  const bf16_t* x;        // packed owned rows   [>= r_pad][512]
  const bf16_t* y;        // packed columns      [>= c_pad][512]
  const float* lse_row;   // [r]
  const float* lse_col;   // [c] or null (c_col = s_col = 0)
  float* slab;            // [n_split][r_pad][512]
  float* ds_part;         // [ceil(r / 64) * n_split]
  int r, c, r_pad, label_off;
  float c_row, c_col, c_diag, s_row, s_col, s_diag;
};
struct BwdFusedBatch {
  BwdFusedProb p[MAX_PROBS];
  int n_probs, n_split, cols_per_split, row_blocks;   // cols_per_split: multiple of 64; row_blocks = max over the problems
  int groups, rows_per_group;                         // set by the launcher: the unit map (clip_bwd.hip)
  int dbg;                                            // debug-switch builds only (MMK_CB_DBG): timing ablations
};
int launch_clip_bwd_fused(const BwdFusedBatch& b, const float* scale, hipStream_t st);

// ------------------------------------------------------------------ forward statistics of row-sharded directions (clip_bwd.hip)
struct FwdShardProb {
  const bf16_t* x;        // packed owned rows   [>= r][512]
  const bf16_t* y;        // packed columns      [>= c][512]
  float2* part;           // [n_split][part_ld]: per (column split, owned row) the partial (ref2, sum of 2^(u - ref2)), log2 domain
  float* diag;            // [r]: the positive logit s * <x_i, y_label(i)>
  int r, c, part_ld, label_off;
};
struct FwdShardBatch {
  FwdShardProb p[MAX_PROBS];
  int n_probs, n_split, cols_per_split, row_blocks;   // as BwdFusedBatch
  int groups, rows_per_group;
};
int launch_clip_fwd_shard(const FwdShardBatch& b, const float* scale, hipStream_t st);

}  // namespace mmk
