// attention of ONE query per (sample, head) against all keys -- the token-0 row of the LAST layer of a tower that is
// pooled at token 0 (mmlearn/modules/encoders/clip.py:463-470: `last_hidden_state[:, 0, :]`; a [CLS]-pooled BERT): fused.py's
// `cls_only_last_layer` computes keys and values for every token and everything after them for token 0 only.  The library's SDPA serves
// that shape with its flash kernels: 1.63 ms forward + backward at B = 1024, H = 12, L = 197 (0.81 ms at L = 77) for a pass whose bytes
// -- K, V read twice, dK, dV written -- move in 0.37 ms.  Here one WAVE owns one (sample, head):
//   forward   lane = key (up to four keys per lane, L <= 256): s_j = <q, k_j> over the lane's own 128-byte key row, softmax over the wave,
//             o = sum_j p_j m_j v_j accumulated per lane over its keys and folded across the lanes by a transpose-reduce (63 exchanges);
//   backward  the same walk: p_j from the saved log-sum-exp, dp_j = m_j <dO, v_j>, ds_j = p_j (dp_j - sum_k p_k dp_k); the lane WRITES its
//             keys' rows dk_j = scale ds_j q and dv_j = p_j m_j dO (128 contiguous bytes each) and folds dq = scale sum_j ds_j k_j.
// m_j = dropout keep factor of (query 0, key j) (attention-probability dropout, HF BERT's 0.1 in training): the counter-based mask of
// csrc/common.h (drop_key / drop_word) that csrc/attention.hip uses, regenerated in the backward.  HBM-bound by construction, no MFMA:
// 2 x 64 FMAs per key and direction.
#include <hip/hip_ext.h>
#include <stdint.h>

#include "common.h"

namespace mmk {

constexpr int CA_DH = 64;        // head dim
constexpr int CA_MAXT = 4;       // keys per lane: L <= 256
constexpr float CA_LOG2E = 1.4426950408889634f;

struct ClsAttnArgs {
  const bf16_t* q;     // [B, H, 64] contiguous
  const bf16_t* k;     // element (b, l, h, d) at k + b * kv_sb + l * kv_sl + h * 64 + d
  const bf16_t* v;
  bf16_t* o;           // [B, H, 64]
  float* lse2;         // [B, H]: max + log2(sum) of the base-2 logits
  const bf16_t* dout;  // [B, H, 64]
  bf16_t* dq;          // [B, H, 64]
  bf16_t* dk;          // strides g_sb, g_sl (same layout rule as k / v)
  bf16_t* dv;
  long kv_sb, kv_sl, g_sb, g_sl;
  int B, H, L;
  float scale;
  uint32_t seed_lo, seed_hi, drop_thr;
  float drop_scale;
  const float* kbias;  // optional key bias records [B][256] (mmk_attn_key_bias): added to the base-2 logit of key j of sample b
};

typedef bf16_t ca_bf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float ca_wave_max(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v = fmaxf(v, __shfl_xor(v, s));
  return v;
}
__device__ __forceinline__ float ca_wave_sum(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
  return v;
}
// acc[d] summed over the 64 lanes; lane d returns the total of column d.  Round with stride S: the lane keeps the half of its
// remaining columns whose index bit log2(S) equals its own lane bit, and sends the other half to lane ^ S.
__device__ __forceinline__ float ca_transpose_reduce(float (&acc)[CA_DH], int lane) {
#define CA_ROUND(N, S)                                          \
  {                                                             \
    const bool up = (lane & (S)) != 0;                          \
    _Pragma("unroll") for (int i = 0; i < (N) / 2; ++i) {       \
      const float send = up ? acc[i] : acc[(N) / 2 + i];        \
      const float keep = up ? acc[(N) / 2 + i] : acc[i];        \
      acc[i] = keep + __shfl_xor(send, (S));                    \
    }                                                           \
  }
  CA_ROUND(64, 32) CA_ROUND(32, 16) CA_ROUND(16, 8) CA_ROUND(8, 4) CA_ROUND(4, 2) CA_ROUND(2, 1)
#undef CA_ROUND
  return acc[0];
}
__device__ __forceinline__ void ca_load_row(const bf16_t* p, ca_bf8 (&r)[8]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) r[c] = *reinterpret_cast<const ca_bf8*>(p + 8 * c);
}
__device__ __forceinline__ float ca_dot(const ca_bf8 (&a)[8], const ca_bf8 (&b)[8]) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      s0 = fmaf((float)a[c][e], (float)b[c][e], s0);
      s1 = fmaf((float)a[c][e + 1], (float)b[c][e + 1], s1);
    }
  return s0 + s1;
}
__device__ __forceinline__ float ca_keep(const ClsAttnArgs& a, uint32_t dkey, int j) {
  if (a.drop_thr == 0u) return 1.f;
  const uint32_t w = drop_word(dkey, 0, j >> 1);
  return ((w >> (16 * (j & 1))) & 0xFFFFu) < a.drop_thr ? 0.f : a.drop_scale;
}

__global__ __launch_bounds__(256) void cls_attn_fwd_kernel(const ClsAttnArgs a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int item = blockIdx.x * 4 + wave;
  if (item >= a.B * a.H) return;
  const int b = item / a.H, hh = item % a.H;
  const bf16_t* kb = a.k + (long)b * a.kv_sb + hh * CA_DH;
  const bf16_t* vb = a.v + (long)b * a.kv_sb + hh * CA_DH;
  ca_bf8 qr[8];
  ca_load_row(a.q + (long)item * CA_DH, qr);
  const float sc2 = a.scale * CA_LOG2E;
  const uint32_t dkey = a.drop_thr ? drop_key(a.seed_lo, a.seed_hi, (uint32_t)item) : 0u;
  float s[CA_MAXT];
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < CA_MAXT; ++t) {
    const int j = lane + 64 * t;
    s[t] = -INFINITY;
    if (j < a.L) {
      ca_bf8 kr[8];
      ca_load_row(kb + (long)j * a.kv_sl, kr);
      s[t] = fmaf(ca_dot(qr, kr), sc2, a.kbias ? a.kbias[(long)b * 256 + j] : 0.f);
    }
    m = fmaxf(m, s[t]);
  }
  m = ca_wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < CA_MAXT; ++t) {
    s[t] = (lane + 64 * t < a.L) ? __builtin_amdgcn_exp2f(s[t] - m) : 0.f;
    sum += s[t];
  }
  sum = ca_wave_sum(sum);
  const float inv = 1.f / sum;
  if (lane == 0) a.lse2[item] = m + __builtin_amdgcn_logf(sum);   // v_log_f32 = log2
  float acc[CA_DH];
#pragma unroll
  for (int d = 0; d < CA_DH; ++d) acc[d] = 0.f;
#pragma unroll
  for (int t = 0; t < CA_MAXT; ++t) {
    const int j = lane + 64 * t;
    if (j < a.L) {
      const float pm = s[t] * inv * ca_keep(a, dkey, j);
      ca_bf8 vr[8];
      ca_load_row(vb + (long)j * a.kv_sl, vr);
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[8 * c + e] = fmaf(pm, (float)vr[c][e], acc[8 * c + e]);
    }
  }
  const float od = ca_transpose_reduce(acc, lane);
  a.o[(long)item * CA_DH + lane] = (bf16_t)od;
}

__global__ __launch_bounds__(256) void cls_attn_bwd_kernel(const ClsAttnArgs a) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int item = blockIdx.x * 4 + wave;
  if (item >= a.B * a.H) return;
  const int b = item / a.H, hh = item % a.H;
  const bf16_t* kb = a.k + (long)b * a.kv_sb + hh * CA_DH;
  const bf16_t* vb = a.v + (long)b * a.kv_sb + hh * CA_DH;
  bf16_t* dkb = a.dk + (long)b * a.g_sb + hh * CA_DH;
  bf16_t* dvb = a.dv + (long)b * a.g_sb + hh * CA_DH;
  ca_bf8 qr[8], gr[8];
  ca_load_row(a.q + (long)item * CA_DH, qr);
  ca_load_row(a.dout + (long)item * CA_DH, gr);
  const float sc2 = a.scale * CA_LOG2E;
  const float l2 = a.lse2[item];
  const uint32_t dkey = a.drop_thr ? drop_key(a.seed_lo, a.seed_hi, (uint32_t)item) : 0u;
  // pass 1: p_j, dp_j, delta = sum_j p_j dp_j (the lane's rows are read again in pass 2: they come back from the L2 / L1)
  float p[CA_MAXT], dp[CA_MAXT], keep[CA_MAXT];
  float del = 0.f;
#pragma unroll
  for (int t = 0; t < CA_MAXT; ++t) {
    const int j = lane + 64 * t;
    p[t] = dp[t] = keep[t] = 0.f;
    if (j < a.L) {
      ca_bf8 kr[8], vr[8];
      ca_load_row(kb + (long)j * a.kv_sl, kr);
      ca_load_row(vb + (long)j * a.kv_sl, vr);
      // (bias first, then the lse: an all-masked row has both at about -1e30, see bwd_prob in attention.hip)
      p[t] = __builtin_amdgcn_exp2f(fmaf(ca_dot(qr, kr), sc2, a.kbias ? a.kbias[(long)b * 256 + j] : 0.f) - l2);
      keep[t] = ca_keep(a, dkey, j);
      dp[t] = keep[t] * ca_dot(gr, vr);
      del = fmaf(p[t], dp[t], del);
    }
  }
  del = ca_wave_sum(del);
  float acc[CA_DH];
#pragma unroll
  for (int d = 0; d < CA_DH; ++d) acc[d] = 0.f;
#pragma unroll
  for (int t = 0; t < CA_MAXT; ++t) {
    const int j = lane + 64 * t;
    if (j < a.L) {
      const float ds = p[t] * (dp[t] - del) * a.scale;   // d loss / d (q . k_j), the softmax scale folded in
      const float pv = p[t] * keep[t];
      ca_bf8 kr[8];
      ca_load_row(kb + (long)j * a.kv_sl, kr);
      bf16_t* dkr = dkb + (long)j * a.g_sl;
      bf16_t* dvr = dvb + (long)j * a.g_sl;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        ca_bf8 ok, ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          acc[8 * c + e] = fmaf(ds, (float)kr[c][e], acc[8 * c + e]);
          ok[e] = (bf16_t)(ds * (float)qr[c][e]);
          ov[e] = (bf16_t)(pv * (float)gr[c][e]);
        }
        *reinterpret_cast<ca_bf8*>(dkr + 8 * c) = ok;
        *reinterpret_cast<ca_bf8*>(dvr + 8 * c) = ov;
      }
    }
  }
  const float dqd = ca_transpose_reduce(acc, lane);
  a.dq[(long)item * CA_DH + lane] = (bf16_t)dqd;
}

static int cls_attn_check(const ClsAttnArgs& a, int dh) {
  MMK_REQUIRE(dh == CA_DH, "cls_attn: head dim must be 64");
  MMK_REQUIRE(a.B > 0 && a.H > 0 && a.L > 0 && a.L <= 64 * CA_MAXT, "cls_attn: need 1 <= L <= 256");
  MMK_REQUIRE(a.kv_sl % 8 == 0 && a.kv_sb % 8 == 0, "cls_attn: key / value rows must be 16-byte aligned");
  MMK_REQUIRE((long)a.B * a.H < (1l << 31) - 4, "cls_attn: too many (sample, head) items");
  return 0;
}

}  // namespace mmk

using namespace mmk;

extern "C" {

int mmk_cls_attn_supported(int L, int dh) { return dh == CA_DH && L >= 1 && L <= 64 * CA_MAXT; }

int mmk_cls_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse2, int B, int H, int L, int dh, int64_t kv_sb,
                     int64_t kv_sl, float scale, float dropout_p, uint64_t seed, const float* key_bias, void* stream) {
  MMK_REQUIRE(q && k && v && o && lse2, "cls_attn_fwd: bad arguments");
  ClsAttnArgs a = {};
  a.q = static_cast<const bf16_t*>(q); a.k = static_cast<const bf16_t*>(k); a.v = static_cast<const bf16_t*>(v);
  a.o = static_cast<bf16_t*>(o); a.lse2 = lse2; a.kbias = key_bias;
  a.B = B; a.H = H; a.L = L; a.kv_sb = kv_sb; a.kv_sl = kv_sl; a.scale = scale;
  if (int rc = cls_attn_check(a, dh)) return rc;
  MMK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "cls_attn: dropout probability must be in [0, 1)");
  drop_params(dropout_p, seed, &a.seed_lo, &a.seed_hi, &a.drop_thr, &a.drop_scale);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)(((long)B * H + 3) / 4);
  {
    ProfEvents pe(MMK_K_ATTN_FWD);
    hipExtLaunchKernelGGL(cls_attn_fwd_kernel, dim3(grid), dim3(256), 0, st, pe.start, pe.stop, 0, a);
  }
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_cls_attn_bwd(const void* q, const void* k, const void* v, const void* dout, const float* lse2, void* dq, void* dk, void* dv, int B,
                     int H, int L, int dh, int64_t kv_sb, int64_t kv_sl, int64_t g_sb, int64_t g_sl, float scale, float dropout_p,
                     uint64_t seed, const float* key_bias, void* stream) {
  MMK_REQUIRE(q && k && v && dout && lse2 && dq && dk && dv, "cls_attn_bwd: bad arguments");
  ClsAttnArgs a = {};
  a.q = static_cast<const bf16_t*>(q); a.k = static_cast<const bf16_t*>(k); a.v = static_cast<const bf16_t*>(v);
  a.dout = static_cast<const bf16_t*>(dout); a.lse2 = const_cast<float*>(lse2);
  a.dq = static_cast<bf16_t*>(dq); a.dk = static_cast<bf16_t*>(dk); a.dv = static_cast<bf16_t*>(dv); a.kbias = key_bias;
  a.B = B; a.H = H; a.L = L; a.kv_sb = kv_sb; a.kv_sl = kv_sl; a.g_sb = g_sb; a.g_sl = g_sl; a.scale = scale;
  if (int rc = cls_attn_check(a, dh)) return rc;
  MMK_REQUIRE(g_sl % 8 == 0 && g_sb % 8 == 0, "cls_attn_bwd: gradient rows must be 16-byte aligned");
  MMK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "cls_attn: dropout probability must be in [0, 1)");
  drop_params(dropout_p, seed, &a.seed_lo, &a.seed_hi, &a.drop_thr, &a.drop_scale);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)(((long)B * H + 3) / 4);
  {
    ProfEvents pe(MMK_K_ATTN_BWD);
    hipExtLaunchKernelGGL(cls_attn_bwd_kernel, dim3(grid), dim3(256), 0, st, pe.start, pe.stop, 0, a);
  }
  MMK_LAUNCH_CHECK();
  return 0;
}
}
