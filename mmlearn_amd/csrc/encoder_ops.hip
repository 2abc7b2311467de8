// SURVEY 8(f1) -- first widening into the encoder step: the HBM-bound row / elementwise ops that dominate the
// non-GEMM time of the ViT-B/16 + BERT-base step under bf16 autocast (rocprof: LayerNorm fwd+bwd ~126 ms and the
// unfused quick-GELU chain ~100 ms of a 425 ms step).  Replaces, inside the encoders the reference instantiates
// (mmlearn/modules/encoders/clip.py, text.py, vision.py -> torch.nn.LayerNorm / HF activations):
//   * F.layer_norm forward + backward (3 ATen kernels in backward) -> one forward kernel, one fused backward kernel
//     (dx + per-block partial dgamma/dbeta) + a tiny column reduce;
//   * x * sigmoid(1.702 x)  (3 elementwise kernels forward, ~6 backward) -> one kernel each way.
// All are "one wave per row" / 16-byte-per-lane streaming kernels: coalesced HBM traffic, wave-shuffle reductions.
#include <algorithm>

#include "common.h"

namespace mmk {

constexpr int LN_ROWS_PER_WAVE = 16;  // rows each wave walks in the backward kernel (64 rows per block partial)
constexpr int LN_MAX_VEC = 8;        // float4 per lane held in registers: D <= 64*4*8 = 2048 (VEC = 4 when D <= 1024)

// ------------------------------------------------------------------ LayerNorm forward
// y = (x - mean) * rstd * w + b, statistics in f32 over the row held in registers (two-pass variance).
template <typename X, typename Y, int VEC>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const X* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, Y* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out, long rows,
                                                            int d, float eps) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const X* xr = x + row * d;
  Y* yr = y + row * d;
  float4 v[VEC];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const int c = (lane + 64 * k) * 4;
    if (c < d) {
      v[k] = Vec4<X>::load(xr + c);
      s += v[k].x + v[k].y + v[k].z + v[k].w;
    }
  }
  const float mean = wave_sum(s) / d;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const int c = (lane + 64 * k) * 4;
    if (c < d) {
      const float a0 = v[k].x - mean, a1 = v[k].y - mean, a2 = v[k].z - mean, a3 = v[k].w - mean;
      ss += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / d + eps);
  if (lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const int c = (lane + 64 * k) * 4;
    if (c < d) {
      float4 g = make_float4(1.f, 1.f, 1.f, 1.f), o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (w) g = *reinterpret_cast<const float4*>(w + c);
      if (b) o = *reinterpret_cast<const float4*>(b + c);
      float4 r;
      r.x = (v[k].x - mean) * rstd * g.x + o.x;
      r.y = (v[k].y - mean) * rstd * g.y + o.y;
      r.z = (v[k].z - mean) * rstd * g.z + o.z;
      r.w = (v[k].w - mean) * rstd * g.w + o.w;
      Vec4<Y>::store(yr + c, r);
    }
  }
}

// ------------------------------------------------------------------ LayerNorm backward (fused)
// per row: xhat = (x-mean)*rstd, gy = dy*w, dx = rstd*(gy - mean(gy) - xhat*mean(gy*xhat));
// per block: partial dgamma = sum dy*xhat, dbeta = sum dy over its rows -> part[block][2][d]
template <typename X, typename G, int VEC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const X* __restrict__ x, const G* __restrict__ dy,
                                                            const float* __restrict__ w, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, X* __restrict__ dx,
                                                            float* __restrict__ part, long rows, int d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [4 waves][2][d]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 dg[VEC], db[VEC], wv[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    dg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = (lane + 64 * k) * 4;
    wv[k] = (w && c < d) ? *reinterpret_cast<const float4*>(w + c) : make_float4(1.f, 1.f, 1.f, 1.f);
  }
  const long row0 = ((long)blockIdx.x * 4 + wave) * LN_ROWS_PER_WAVE;
  for (int q = 0; q < LN_ROWS_PER_WAVE; ++q) {
    const long row = row0 + q;
    if (row >= rows) break;
    const X* xr = x + row * d;
    const G* gr = dy + row * d;
    const float mean = mean_in[row], rstd = rstd_in[row];
    float4 xh[VEC], gy[VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int c = (lane + 64 * k) * 4;
      if (c < d) {
        const float4 xv = Vec4<X>::load(xr + c);
        const float4 g = Vec4<G>::load(gr + c);
        xh[k] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        gy[k] = make_float4(g.x * wv[k].x, g.y * wv[k].y, g.z * wv[k].z, g.w * wv[k].w);
        s1 += gy[k].x + gy[k].y + gy[k].z + gy[k].w;
        s2 += gy[k].x * xh[k].x + gy[k].y * xh[k].y + gy[k].z * xh[k].z + gy[k].w * xh[k].w;
        dg[k].x += g.x * xh[k].x; dg[k].y += g.y * xh[k].y; dg[k].z += g.z * xh[k].z; dg[k].w += g.w * xh[k].w;
        db[k].x += g.x; db[k].y += g.y; db[k].z += g.z; db[k].w += g.w;
      }
    }
    const float m1 = wave_sum(s1) / d, m2 = wave_sum(s2) / d;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int c = (lane + 64 * k) * 4;
      if (c < d) {
        float4 r;
        r.x = rstd * (gy[k].x - m1 - xh[k].x * m2);
        r.y = rstd * (gy[k].y - m1 - xh[k].y * m2);
        r.z = rstd * (gy[k].z - m1 - xh[k].z * m2);
        r.w = rstd * (gy[k].w - m1 - xh[k].w * m2);
        Vec4<X>::store(dx + row * d + c, r);
      }
    }
  }
  if (part == nullptr) return;
  // combine the 4 waves' column partials through LDS, one [2][d] slab per block
  float* mine = lds + (size_t)wave * 2 * d;
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const int c = (lane + 64 * k) * 4;
    if (c < d) {
      *reinterpret_cast<float4*>(mine + c) = dg[k];
      *reinterpret_cast<float4*>(mine + d + c) = db[k];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * d; c += 256)
    part[(size_t)blockIdx.x * 2 * d + c] = lds[c] + lds[2 * d + c] + lds[4 * d + c] + lds[6 * d + c];
}

// ------------------------------------------------------------------ LayerNorm of narrow rows (d <= 128): two rows per wave
// With one row per wave a 96-wide row (HTSAT's first resolution: 1 M rows per LayerNorm) keeps 24 of 64 lanes busy and the kernels run
// at 2 TB/s.  Here a row takes 32 lanes x float4: lanes 0-31 / 32-63 hold two consecutive rows, the reductions stay inside a half
// (xor 16 .. 1).  Same arithmetic as the wide kernels (two-pass variance in registers); the backward keeps the wide kernel's
// block -> rows map (64 rows per block partial), so the callers' workspace sizes do not change.
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename X, typename Y>
__global__ __launch_bounds__(256) void layernorm_fwd_narrow_kernel(const X* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ b, Y* __restrict__ y,
                                                                   float* __restrict__ mean_out, float* __restrict__ rstd_out, long rows,
                                                                   int d, float eps) {
  const int lane = threadIdx.x & 63, sub = lane & 31, c = sub * 4;
  const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 6) * 2 + (lane >> 5);
  const bool on = row < rows && c < d;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (on) v = Vec4<X>::load(x + row * d + c);
  const float mean = half_sum(v.x + v.y + v.z + v.w) / d;
  const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
  const float rstd = rsqrtf(half_sum(on ? a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3 : 0.f) / d + eps);
  if (sub == 0 && row < rows) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
  if (on) {
    float4 g = make_float4(1.f, 1.f, 1.f, 1.f), o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (w) g = *reinterpret_cast<const float4*>(w + c);
    if (b) o = *reinterpret_cast<const float4*>(b + c);
    Vec4<Y>::store(y + row * d + c, make_float4(a0 * rstd * g.x + o.x, a1 * rstd * g.y + o.y, a2 * rstd * g.z + o.z, a3 * rstd * g.w + o.w));
  }
}

template <typename X, typename G>
__global__ __launch_bounds__(256) void layernorm_bwd_narrow_kernel(const X* __restrict__ x, const G* __restrict__ dy, const float* __restrict__ w,
                                                                   const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                                   X* __restrict__ dx, float* __restrict__ part, long rows, int d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [4 waves x 2 halves][2][d]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane & 31, seg = lane >> 5, c = sub * 4;
  float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
  const float4 wv = (w && c < d) ? *reinterpret_cast<const float4*>(w + c) : make_float4(1.f, 1.f, 1.f, 1.f);
  const long row0 = ((long)blockIdx.x * 4 + wave) * LN_ROWS_PER_WAVE;
  for (int q = 0; q < LN_ROWS_PER_WAVE / 2; ++q) {
    const long row = row0 + 2 * q + seg;
    const bool on = row < rows && c < d;
    float4 xh = make_float4(0.f, 0.f, 0.f, 0.f), gy = xh;
    float rstd = 0.f;
    if (on) {
      const float mean = mean_in[row];
      rstd = rstd_in[row];
      const float4 xv = Vec4<X>::load(x + row * d + c);
      const float4 g = Vec4<G>::load(dy + row * d + c);
      xh = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
      gy = make_float4(g.x * wv.x, g.y * wv.y, g.z * wv.z, g.w * wv.w);
      dg.x += g.x * xh.x; dg.y += g.y * xh.y; dg.z += g.z * xh.z; dg.w += g.w * xh.w;
      db.x += g.x; db.y += g.y; db.z += g.z; db.w += g.w;
    }
    const float m1 = half_sum(gy.x + gy.y + gy.z + gy.w) / d;
    const float m2 = half_sum(gy.x * xh.x + gy.y * xh.y + gy.z * xh.z + gy.w * xh.w) / d;
    if (on)
      Vec4<X>::store(dx + row * d + c, make_float4(rstd * (gy.x - m1 - xh.x * m2), rstd * (gy.y - m1 - xh.y * m2),
                                                  rstd * (gy.z - m1 - xh.z * m2), rstd * (gy.w - m1 - xh.w * m2)));
  }
  if (part == nullptr) return;
  float* mine = lds + (size_t)(wave * 2 + seg) * 2 * d;
  if (c < d) {
    *reinterpret_cast<float4*>(mine + c) = dg;
    *reinterpret_cast<float4*>(mine + d + c) = db;
  }
  __syncthreads();
  for (int cc = threadIdx.x; cc < 2 * d; cc += 256) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += lds[(size_t)j * 2 * d + cc];
    part[(size_t)blockIdx.x * 2 * d + cc] = t;
  }
}

// ------------------------------------------------------------------ residual add (+ dropout) + LayerNorm, fused
// s = r + dropout(x);  y = LN(s).   The transformer blocks' "hidden = residual + sublayer(...)" followed by the next
// LayerNorm (HF CLIPEncoderLayer: residual + attn -> layer_norm2; BertSelfOutput / BertOutput: LayerNorm(dropout(dense)
// + input)).  One pass instead of add kernel + LayerNorm kernel (+ dropout kernel): reads x and r, writes s (only when
// the caller needs the sum, i.e. pre-LN blocks) and y.  Dropout uses the counter-based mask of common.h keyed by
// (seed, row): the backward regenerates it.
template <typename XT, typename Y, int VEC, bool DROP>
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const XT* __restrict__ x, const float* __restrict__ xb,
                                                                const float* __restrict__ r, const float* __restrict__ w,
                                                                const float* __restrict__ b,
                                                                float* __restrict__ s_out, Y* __restrict__ y,
                                                                bf16_t* __restrict__ y_twin, float* __restrict__ mean_out,
                                                                float* __restrict__ rstd_out,
                                                                long rows, int d, float eps, uint32_t seed_lo, uint32_t seed_hi,
                                                                uint32_t drop_thr, float drop_scale) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const uint32_t key = DROP ? drop_key(seed_lo, seed_hi, (uint32_t)row) : 0u;
  float4 v[VEC];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const int c = (lane + 64 * k) * 4;
    if (c < d) {
      float4 xv = Vec4<XT>::load(x + row * d + c);
      if (xb) {  // the producing Linear's bias, deferred into this kernel (its gradient comes out of the backward)
        const float4 bb = *reinterpret_cast<const float4*>(xb + c);
        xv.x += bb.x; xv.y += bb.y; xv.z += bb.z; xv.w += bb.w;
      }
      if (DROP) {
        const uint32_t w0 = drop_word(key, 0, c >> 1), w1 = drop_word(key, 0, (c >> 1) + 1);
        xv.x = (w0 & 0xFFFFu) < drop_thr ? 0.f : xv.x * drop_scale;
        xv.y = (w0 >> 16) < drop_thr ? 0.f : xv.y * drop_scale;
        xv.z = (w1 & 0xFFFFu) < drop_thr ? 0.f : xv.z * drop_scale;
        xv.w = (w1 >> 16) < drop_thr ? 0.f : xv.w * drop_scale;
      }
      const float4 rv = *reinterpret_cast<const float4*>(r + row * d + c);
      v[k] = make_float4(xv.x + rv.x, xv.y + rv.y, xv.z + rv.z, xv.w + rv.w);
      if (s_out) *reinterpret_cast<float4*>(s_out + row * d + c) = v[k];
      sum += v[k].x + v[k].y + v[k].z + v[k].w;
    }
  }
  const float mean = wave_sum(sum) / d;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const int c = (lane + 64 * k) * 4;
    if (c < d) {
      const float a0 = v[k].x - mean, a1 = v[k].y - mean, a2 = v[k].z - mean, a3 = v[k].w - mean;
      ss += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / d + eps);
  if (lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const int c = (lane + 64 * k) * 4;
    if (c < d) {
      float4 g = make_float4(1.f, 1.f, 1.f, 1.f), o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (w) g = *reinterpret_cast<const float4*>(w + c);
      if (b) o = *reinterpret_cast<const float4*>(b + c);
      float4 q;
      q.x = (v[k].x - mean) * rstd * g.x + o.x;
      q.y = (v[k].y - mean) * rstd * g.y + o.y;
      q.z = (v[k].z - mean) * rstd * g.z + o.z;
      q.w = (v[k].w - mean) * rstd * g.w + o.w;
      Vec4<Y>::store(y + row * d + c, q);
      if (y_twin) Vec4<bf16_t>::store(y_twin + row * d + c, q);  // bf16 copy for the consumer GEMM (post-LN forks)
    }
  }
}

// Backward of the fused op from the saved sum s:  ds = ds_in + LNbwd(dy)  (ds_in = gradient reaching s through the
// residual stream, absent in post-LN blocks);  dr = ds (f32),  dx = dropout_mask(ds) cast to the sublayer output's dtype.
// Replaces LayerNorm backward + the gradient-accumulation add + the f32 -> bf16 cast of the sublayer gradient.
template <typename G, typename XT, int VEC, bool DROP, bool XBIAS>
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(const float* __restrict__ sv, const G* __restrict__ dy,
                                                                const bf16_t* __restrict__ dy_twin,
                                                                const float* __restrict__ ds_in, const float* __restrict__ w,
                                                                const float* __restrict__ mean_in,
                                                                const float* __restrict__ rstd_in, float* __restrict__ dr,
                                                                XT* __restrict__ dx, float* __restrict__ part, long rows, int d,
                                                                uint32_t seed_lo, uint32_t seed_hi, uint32_t drop_thr,
                                                                float drop_scale) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [4 waves][NP][d], NP = 2 (+1 with XBIAS)
  constexpr int NP = XBIAS ? 3 : 2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 dg[VEC], db[VEC], wv[VEC], dxb[XBIAS ? VEC : 1];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    dg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (XBIAS) dxb[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = (lane + 64 * k) * 4;
    wv[k] = (w && c < d) ? *reinterpret_cast<const float4*>(w + c) : make_float4(1.f, 1.f, 1.f, 1.f);
  }
  const long row0 = ((long)blockIdx.x * 4 + wave) * LN_ROWS_PER_WAVE;
  for (int q = 0; q < LN_ROWS_PER_WAVE; ++q) {
    const long row = row0 + q;
    if (row >= rows) break;
    const float mean = mean_in[row], rstd = rstd_in[row];
    float4 xh[VEC], gy[VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int c = (lane + 64 * k) * 4;
      if (c < d) {
        const float4 xv = *reinterpret_cast<const float4*>(sv + row * d + c);
        float4 g = dy ? Vec4<G>::load(dy + row * d + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (dy_twin) {  // gradient that reached the bf16 twin of y (the consumer GEMM's dX): summed here, not by autograd
          const float4 g2 = Vec4<bf16_t>::load(dy_twin + row * d + c);
          g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
        }
        xh[k] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        gy[k] = make_float4(g.x * wv[k].x, g.y * wv[k].y, g.z * wv[k].z, g.w * wv[k].w);
        s1 += gy[k].x + gy[k].y + gy[k].z + gy[k].w;
        s2 += gy[k].x * xh[k].x + gy[k].y * xh[k].y + gy[k].z * xh[k].z + gy[k].w * xh[k].w;
        dg[k].x += g.x * xh[k].x; dg[k].y += g.y * xh[k].y; dg[k].z += g.z * xh[k].z; dg[k].w += g.w * xh[k].w;
        db[k].x += g.x; db[k].y += g.y; db[k].z += g.z; db[k].w += g.w;
      }
    }
    const float m1 = wave_sum(s1) / d, m2 = wave_sum(s2) / d;
    const uint32_t key = DROP ? drop_key(seed_lo, seed_hi, (uint32_t)row) : 0u;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int c = (lane + 64 * k) * 4;
      if (c < d) {
        float4 t;
        t.x = rstd * (gy[k].x - m1 - xh[k].x * m2);
        t.y = rstd * (gy[k].y - m1 - xh[k].y * m2);
        t.z = rstd * (gy[k].z - m1 - xh[k].z * m2);
        t.w = rstd * (gy[k].w - m1 - xh[k].w * m2);
        if (ds_in) {
          const float4 u = *reinterpret_cast<const float4*>(ds_in + row * d + c);
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        *reinterpret_cast<float4*>(dr + row * d + c) = t;
        if (DROP) {
          const uint32_t w0 = drop_word(key, 0, c >> 1), w1 = drop_word(key, 0, (c >> 1) + 1);
          t.x = (w0 & 0xFFFFu) < drop_thr ? 0.f : t.x * drop_scale;
          t.y = (w0 >> 16) < drop_thr ? 0.f : t.y * drop_scale;
          t.z = (w1 & 0xFFFFu) < drop_thr ? 0.f : t.z * drop_scale;
          t.w = (w1 >> 16) < drop_thr ? 0.f : t.w * drop_scale;
        }
        Vec4<XT>::store(dx + row * d + c, t);
        if (XBIAS) {  // d(bias of the producing Linear) = column sums of dx
          dxb[k].x += (float)(XT)t.x; dxb[k].y += (float)(XT)t.y; dxb[k].z += (float)(XT)t.z; dxb[k].w += (float)(XT)t.w;
        }
      }
    }
  }
  if (part == nullptr) return;
  float* mine = lds + (size_t)wave * NP * d;
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const int c = (lane + 64 * k) * 4;
    if (c < d) {
      *reinterpret_cast<float4*>(mine + c) = dg[k];
      *reinterpret_cast<float4*>(mine + d + c) = db[k];
      if (XBIAS) *reinterpret_cast<float4*>(mine + 2 * d + c) = dxb[k];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < NP * d; c += 256)
    part[(size_t)blockIdx.x * NP * d + c] = lds[c] + lds[NP * d + c] + lds[2 * NP * d + c] + lds[3 * NP * d + c];
}

// ------------------------------------------------------------------ bias + activation, fused (fc1 of the MLPs)
// y = act(x + b) with the producing Linear run WITHOUT its bias: forward is one pass; the backward recomputes x + b,
// writes dx = act'(x + b) * dy and accumulates per-block column sums of dx, which ARE the bias gradient -- the separate
// dY.sum(0) pass of the Linear's backward disappears.  ACT 0: x * sigmoid(1.702 x) (HF quick_gelu), 1: erf GELU.
template <int ACT>
__device__ __forceinline__ float act_fwd(float z) {
  if (ACT == 0) return z / (1.f + __expf(-1.702f * z));
  return 0.5f * z * (1.f + erff(z * 0.70710678118654752f));
}
template <int ACT>
__device__ __forceinline__ float act_grad(float z, float g) {
  if (ACT == 0) {
    const float sg = 1.f / (1.f + __expf(-1.702f * z));
    return g * sg * (1.f + 1.702f * z * (1.f - sg));
  }
  return g * (0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z));
}
constexpr int BA_ROWS = 32;  // rows per block of the backward (one [d] partial per block; 128 rows per block made the kernel 6 % slower)
// 8 elements (16 bytes of bf16) per thread and access: d % 8 == 0
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bias_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ b, T* __restrict__ y,
                                                           long rows, int d) {
  const int d8 = d / 8;
  const long n8 = rows * d8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const int c = (int)(i % d8) * 8;
    float4 v0, v1;
    Vec8<T>::load(x + i * 8, v0, v1);
    const float4 b0 = *reinterpret_cast<const float4*>(b + c), b1 = *reinterpret_cast<const float4*>(b + c + 4);
    v0.x = act_fwd<ACT>(v0.x + b0.x); v0.y = act_fwd<ACT>(v0.y + b0.y); v0.z = act_fwd<ACT>(v0.z + b0.z); v0.w = act_fwd<ACT>(v0.w + b0.w);
    v1.x = act_fwd<ACT>(v1.x + b1.x); v1.y = act_fwd<ACT>(v1.y + b1.y); v1.z = act_fwd<ACT>(v1.z + b1.z); v1.w = act_fwd<ACT>(v1.w + b1.w);
    Vec8<T>::store(y + i * 8, v0, v1);
  }
}
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const T* __restrict__ x, const float* __restrict__ b,
                                                           const T* __restrict__ dy, T* __restrict__ dx, float* __restrict__ part,
                                                           long rows, int d) {
  const int d8 = d / 8;
  const long r0 = (long)blockIdx.x * BA_ROWS, r1 = min(rows, r0 + BA_ROWS);
  for (int cv = threadIdx.x; cv < d8; cv += 256) {  // this thread's column vectors
    const int c = cv * 8;
    const float4 b0 = *reinterpret_cast<const float4*>(b + c), b1 = *reinterpret_cast<const float4*>(b + c + 4);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll 4
    for (long row = r0; row < r1; ++row) {
      float4 v0, v1, g0, g1;
      Vec8<T>::load(x + row * d + c, v0, v1);
      Vec8<T>::load(dy + row * d + c, g0, g1);
      const float4 t0 = make_float4(act_grad<ACT>(v0.x + b0.x, g0.x), act_grad<ACT>(v0.y + b0.y, g0.y), act_grad<ACT>(v0.z + b0.z, g0.z),
                                    act_grad<ACT>(v0.w + b0.w, g0.w));
      const float4 t1 = make_float4(act_grad<ACT>(v1.x + b1.x, g1.x), act_grad<ACT>(v1.y + b1.y, g1.y), act_grad<ACT>(v1.z + b1.z, g1.z),
                                    act_grad<ACT>(v1.w + b1.w, g1.w));
      Vec8<T>::store(dx + row * d + c, t0, t1);
      // the sums are taken over the ROUNDED dx values the weight-gradient GEMM will see
      a0.x += (float)(T)t0.x; a0.y += (float)(T)t0.y; a0.z += (float)(T)t0.z; a0.w += (float)(T)t0.w;
      a1.x += (float)(T)t1.x; a1.y += (float)(T)t1.y; a1.z += (float)(T)t1.z; a1.w += (float)(T)t1.w;
    }
    *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * d + c) = a0;
    *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * d + c + 4) = a1;
  }
}

// dgamma/dbeta = column sums of the block partials.  Stage 1: grid (column chunks of 1024) x (up to 256 row slices), a thread
// owns four consecutive columns (16-byte loads) and keeps four rows in flight; stage 2: 64 columns per workgroup, its four
// waves each take a quarter of the slices, LDS combine, slab p written straight to outs[p].  Fixed summation order.
constexpr int COLSUM_SLICES = 256;
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ part, int n_rows, int n_cols,
                                                     float* __restrict__ out, int rows_per_block) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= n_cols) return;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(n_rows, r0 + rows_per_block);
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    const float4 v0 = *reinterpret_cast<const float4*>(part + (size_t)r * n_cols + c);
    const float4 v1 = *reinterpret_cast<const float4*>(part + (size_t)(r + 1) * n_cols + c);
    const float4 v2 = *reinterpret_cast<const float4*>(part + (size_t)(r + 2) * n_cols + c);
    const float4 v3 = *reinterpret_cast<const float4*>(part + (size_t)(r + 3) * n_cols + c);
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
    a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
  }
  for (; r < r1; ++r) {
    const float4 v0 = *reinterpret_cast<const float4*>(part + (size_t)r * n_cols + c);
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
  }
  *reinterpret_cast<float4*>(out + (size_t)blockIdx.y * n_cols + c) =
      make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
}

// second stage: column sums of the `n_rows` slice rows of a [n_rows][np * d] buffer, slab p written straight to outs[p]
struct ColsumOuts {
  float* p[3];
};
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int n_rows, int d, int np, ColsumOuts outs) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, n_cols = np * d;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < n_cols) {
    int r = grp;
    for (; r + 12 < n_rows; r += 16) {
      s0 += part[(size_t)r * n_cols + c];
      s1 += part[(size_t)(r + 4) * n_cols + c];
      s2 += part[(size_t)(r + 8) * n_cols + c];
      s3 += part[(size_t)(r + 12) * n_cols + c];
    }
    for (; r < n_rows; r += 4) s0 += part[(size_t)r * n_cols + c];
  }
  red[grp][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && c < n_cols) outs.p[c / d][c % d] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// ------------------------------------------------------------------ quick-GELU (x * sigmoid(1.702 x))
template <typename T>
__global__ __launch_bounds__(256) void quick_gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 v = Vec4<T>::load(x + i * 4);
    v.x = v.x / (1.f + __expf(-1.702f * v.x));
    v.y = v.y / (1.f + __expf(-1.702f * v.y));
    v.z = v.z / (1.f + __expf(-1.702f * v.z));
    v.w = v.w / (1.f + __expf(-1.702f * v.w));
    Vec4<T>::store(y + i * 4, v);
  }
}
__device__ __forceinline__ float quick_gelu_grad(float x, float g) {
  const float s = 1.f / (1.f + __expf(-1.702f * x));
  return g * s * (1.f + 1.702f * x * (1.f - s));
}
template <typename T>
__global__ __launch_bounds__(256) void quick_gelu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                             long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = Vec4<T>::load(x + i * 4), g = Vec4<T>::load(dy + i * 4);
    Vec4<T>::store(dx + i * 4, make_float4(quick_gelu_grad(v.x, g.x), quick_gelu_grad(v.y, g.y), quick_gelu_grad(v.z, g.z),
                                           quick_gelu_grad(v.w, g.w)));
  }
}

// ------------------------------------------------------------------ patchify (non-overlapping conv as a GEMM)
// out[(b, py, px)][(c, i, j)] = in[b][c][py P + i][px P + j], cast to bf16: the im2col of a Conv2d whose stride equals its
// kernel (ViT patch embedding) is a pure permutation, so the convolution becomes one [B gh gw, C P P] x [C P P, E] GEMM
// (MIOpen's implicit-GEMM solvers + layout transposes took 5.7 ms per step for it, forward + weight gradient).
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const T* __restrict__ in, bf16_t* __restrict__ out, int B, int C, int H, int W,
                                                       int P) {
  const int gh = H / P, gw = W / P, cpp4 = C * P * P / 4, p4 = P / 4;
  const long n4 = (long)B * gh * gw * cpp4;
  for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < n4; o += (long)gridDim.x * 256) {
    const int col4 = (int)(o % cpp4);
    const long row = o / cpp4;
    const int j4 = col4 % p4, i = (col4 / p4) % P, c = col4 / (p4 * P);
    const int px = (int)(row % gw), py = (int)((row / gw) % gh);
    const long b = row / ((long)gw * gh);
    const float4 v = Vec4<T>::load(in + ((b * C + c) * H + (long)py * P + i) * W + (long)px * P + j4 * 4);
    Vec4<bf16_t>::store(out + o * 4, v);
  }
}

// backward of patchify with respect to the image: the inverse permutation (col2im of a non-overlapping convolution has no sums).
// din[b][c][py P + i][px P + j] = dcols[(b, py, px)][(c, i, j)]; dcols bf16 / f32, din in the image's dtype.
template <typename G, typename T>
__global__ __launch_bounds__(256) void unpatchify_kernel(const G* __restrict__ dcols, T* __restrict__ din, int B, int C, int H, int W, int P) {
  const int gh = H / P, gw = W / P, cpp4 = C * P * P / 4, p4 = P / 4;
  const long n4 = (long)B * gh * gw * cpp4;
  for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < n4; o += (long)gridDim.x * 256) {
    const int col4 = (int)(o % cpp4);
    const long row = o / cpp4;
    const int j4 = col4 % p4, i = (col4 / p4) % P, c = col4 / (p4 * P);
    const int px = (int)(row % gw), py = (int)((row / gw) % gh);
    const long b = row / ((long)gw * gh);
    Vec4<T>::store(din + ((b * C + c) * H + (long)py * P + i) * W + (long)px * P + j4 * 4, Vec4<G>::load(dcols + o * 4));
  }
}

// ------------------------------------------------------------------ weight cast + transposed twin
// W [n, k] (the f32 master weight of an nn.Linear, or bf16 / f16) -> W16 [n, k] bf16, the forward's operand (y = x W16^T), and
// W16T [k, n] bf16 for the backward: dX = dY W is then F.linear(dY, W16T), the operand layout the library's forward kernels
// are built for (54-87 us faster than dY @ W16 on [201728 x 2304] . [2304 x 768], 60 us on the 3072-wide pair; HISTORY.md 5.5).
// One pass over W replaces the per-step autocast cast; 64 x 64 tiles through LDS.
constexpr int CT_TILE = 64;
template <typename T>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const T* __restrict__ w, bf16_t* __restrict__ w16, bf16_t* __restrict__ w16t,
                                                             int n, int k) {
  __shared__ bf16_t tile[CT_TILE][CT_TILE + 2];
  const int k0 = blockIdx.x * CT_TILE, n0 = blockIdx.y * CT_TILE;
  const int c4 = (threadIdx.x & 15) * 4, r = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r + 16 * i;
    if (n0 + row < n && k0 + c4 < k) {
      const long off = (long)(n0 + row) * k + k0 + c4;
      const float4 v = Vec4<T>::load(w + off);
      if (w16 != nullptr) Vec4<bf16_t>::store(w16 + off, v);
      tile[row][c4 + 0] = from_f32<bf16_t>(v.x);
      tile[row][c4 + 1] = from_f32<bf16_t>(v.y);
      tile[row][c4 + 2] = from_f32<bf16_t>(v.z);
      tile[row][c4 + 3] = from_f32<bf16_t>(v.w);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kr = r + 16 * i;   // row of the transposed tile = column of W
    if (k0 + kr < k && n0 + c4 < n) {
      typedef bf16_t bf4 __attribute__((ext_vector_type(4)));
      bf4 o;
      o[0] = tile[c4 + 0][kr];
      o[1] = tile[c4 + 1][kr];
      o[2] = tile[c4 + 2][kr];
      o[3] = tile[c4 + 3][kr];
      *reinterpret_cast<bf4*>(w16t + (long)(k0 + kr) * n + n0 + c4) = o;
    }
  }
}

// ------------------------------------------------------------------ embedding backward (scatter-add of rows)
// dW[ids[r]] += dout[r] for an nn.Embedding table (HF BertEmbeddings word / token-type tables).  ATen sorts the ids and
// runs a segmented reduction (~1 ms per table and step here); this kernel gives a wave 16 consecutive rows, sums runs of
// equal ids in registers (token-type ids are all equal, padded sequences repeat the pad id) and flushes each run with one
// hardware float atomic per element.  dW is f32 and zeroed by the caller.
constexpr int EMB_ROWS = 16;
template <typename G>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const G* __restrict__ dout, const int64_t* __restrict__ ids,
                                                            float* __restrict__ dw, long rows, int d, long vocab) {
  const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * EMB_ROWS;
  const int lane = threadIdx.x & 63;
  for (int c0 = lane * 4; c0 < d; c0 += 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    long cur = -1;
    for (int q = 0; q < EMB_ROWS; ++q) {
      const long row = r0 + q;
      if (row >= rows) break;
      const long id = ids[row];
      if (id != cur) {
        if (cur >= 0 && cur < vocab) {
          float* o = dw + cur * d + c0;
          unsafeAtomicAdd(o, acc.x); unsafeAtomicAdd(o + 1, acc.y); unsafeAtomicAdd(o + 2, acc.z); unsafeAtomicAdd(o + 3, acc.w);
        }
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        cur = id;
      }
      const float4 v = Vec4<G>::load(dout + row * d + c0);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (cur >= 0 && cur < vocab) {
      float* o = dw + cur * d + c0;
      unsafeAtomicAdd(o, acc.x); unsafeAtomicAdd(o + 1, acc.y); unsafeAtomicAdd(o + 2, acc.z); unsafeAtomicAdd(o + 3, acc.w);
    }
  }
}

// ------------------------------------------------------------------ bicubic resize along one axis (HTSAT's spectrogram stretch)
// HF ClapAudioEncoder.reshape_mel2img stretches the [B, 1, T = 1001, F = 64] log-mel input to T' = 1024 frames with
// F.interpolate(mode="bicubic", align_corners=True): ATen's 2-D kernel evaluates 4 x 4 taps per output although the F axis keeps
// its length (its four weights are exactly 0, 1, 0, 0) -- 1.3 ms forward and 1.3 ms backward for 67 MB, the gradient being needed
// because a BatchNorm sits in front.  These are the same taps along T only (A = -0.75, source coordinate scale * o with
// scale = (T - 1) / (T' - 1) in float, indices clamped: ATen's arithmetic, same order of the four products), float4 along F.
__device__ __forceinline__ void cubic_taps(float scale, int o, int h_in, int (&idx)[4], float (&c)[4]) {
  // the product is rounded to float BEFORE the fraction is taken, as in ATen's kernel: at o ~ 1000 its spacing is 6e-5, and the fused
  // multiply-subtract the compiler forms here otherwise (v_fma_f32 scale, o, -floor) moves the outputs by 2e-4.  The empty asm makes
  // the rounded product a value the subtraction cannot look through.
  float real = scale * (float)o;
  asm volatile("" : "+v"(real));
  const int i0 = (int)floorf(real);
  const float t = real - (float)i0, A = -0.75f;
  const float x0 = t + 1.f, x2 = 1.f - t, x3 = x2 + 1.f;
  c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  c[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
  c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
#pragma unroll
  for (int k = 0; k < 4; ++k) idx[k] = min(max(i0 - 1 + k, 0), h_in - 1);
}

__global__ __launch_bounds__(256) void cubic_rows_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n_img, int h_in, int h_out,
                                                             int w4, float scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_img * h_out * w4) return;
  const int c = (int)(i % w4), o = (int)((i / w4) % h_out);
  const long n = i / ((long)w4 * h_out);
  int idx[4];
  float cf[4];
  cubic_taps(scale, o, h_in, idx, cf);
  const float4* xs = reinterpret_cast<const float4*>(x) + n * h_in * w4 + c;
  const float4 a = xs[(long)idx[0] * w4], b = xs[(long)idx[1] * w4], d = xs[(long)idx[2] * w4], e = xs[(long)idx[3] * w4];
  float4 r;
  r.x = a.x * cf[0] + b.x * cf[1] + d.x * cf[2] + e.x * cf[3];
  r.y = a.y * cf[0] + b.y * cf[1] + d.y * cf[2] + e.y * cf[3];
  r.z = a.z * cf[0] + b.z * cf[1] + d.z * cf[2] + e.z * cf[3];
  r.w = a.w * cf[0] + b.w * cf[1] + d.w * cf[2] + e.w * cf[3];
  reinterpret_cast<float4*>(y)[i] = r;
}

// gradient in gather form: input row h collects from the (few) output rows one of whose clamped taps is h -- no atomics, fixed order
__global__ __launch_bounds__(256) void cubic_rows_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long n_img, int h_in, int h_out,
                                                             int w4, float scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_img * h_in * w4) return;
  const int c = (int)(i % w4), h = (int)((i / w4) % h_in);
  const long n = i / ((long)w4 * h_in);
  // outputs whose source coordinate lies within [h - 2, h + 2] (+ one of slack each side for the float rounding of scale * o)
  const int o_lo = max(0, (int)floorf((float)(h - 2) / scale) - 1), o_hi = min(h_out - 1, (int)ceilf((float)(h + 2) / scale) + 1);
  const float4* gs = reinterpret_cast<const float4*>(dy) + n * h_out * w4 + c;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int o = o_lo; o <= o_hi; ++o) {
    int idx[4];
    float cf[4];
    cubic_taps(scale, o, h_in, idx, cf);
    float wsum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) wsum += idx[k] == h ? cf[k] : 0.f;
    if (wsum != 0.f) {
      const float4 g = gs[(long)o * w4];
      acc.x += g.x * wsum; acc.y += g.y * wsum; acc.z += g.z * wsum; acc.w += g.w * wsum;
    }
  }
  reinterpret_cast<float4*>(dx)[i] = acc;
}

// ------------------------------------------------------------------ column sums of a [rows, n] bf16 / f32 matrix (bias gradient of a Linear)
// dY.sum(0) for the Linears whose bias no other kernel takes care of (HTSAT's: 73 per step, [1 M x 96] ... [16 k x 3072] bf16).  The block is
// laid out as (256 / cw) rows x cw 16-byte chunks, cw = chunks of the column group (<= 256 columns of bf16 x 8): every load instruction of
// a wave covers whole rows back to back, whatever n is (96 columns = 12 chunks -> 21 rows per pass), rows are unrolled four deep, and the
// threads of one chunk column are added up through LDS.  One f32 partial row per (slice, column group); colsum_final_kernel finishes.
constexpr int CSR_MAX_SLICES = 1024;
template <typename T>
__global__ __launch_bounds__(256) void colsum_rows_kernel(const T* __restrict__ x, long rows, int n, float* __restrict__ part, long rows_per_slice) {
  constexpr int EPC = 16 / sizeof(T);              // elements per 16-byte chunk
  __shared__ float red[256][EPC + 1];
  const int cpr = n / EPC;                         // chunks per row
  const int c0 = blockIdx.x * 256, cw = min(256, cpr - c0);
  const int rpp = 256 / cw;                        // rows per pass of the block
  const int t = threadIdx.x, tr = t / cw, tc = t - tr * cw;
  const long r0 = (long)blockIdx.y * rows_per_slice, r1 = min(rows, r0 + rows_per_slice);
  float acc[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
  if (tr < rpp) {
    const T* base = x + (long)(c0 + tc) * EPC;
    long r = r0 + tr;
    for (; r + 3L * rpp < r1; r += 4L * rpp) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(base + (r + (long)u * rpp) * n);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (sizeof(T) == 4) {
          acc[0] += v[u].x; acc[1] += v[u].y; acc[2] += v[u].z; acc[3] += v[u].w;
        } else {
          const uint32_t w[4] = {__float_as_uint(v[u].x), __float_as_uint(v[u].y), __float_as_uint(v[u].z), __float_as_uint(v[u].w)};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[(2 * q) % EPC] += __uint_as_float(w[q] << 16);
            acc[(2 * q + 1) % EPC] += __uint_as_float(w[q] & 0xffff0000u);
          }
        }
      }
    }
    for (; r < r1; r += rpp) {
      const float4 v = *reinterpret_cast<const float4*>(base + r * n);
      if (sizeof(T) == 4) {
        acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
      } else {
        const uint32_t w[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[(2 * q) % EPC] += __uint_as_float(w[q] << 16);
          acc[(2 * q + 1) % EPC] += __uint_as_float(w[q] & 0xffff0000u);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) red[t][e] = acc[e];
  __syncthreads();
  if (t < cw) {
    float s[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = 0.f;
    for (int q = 0; q < rpp; ++q)
#pragma unroll
      for (int e = 0; e < EPC; ++e) s[e] += red[q * cw + t][e];
#pragma unroll
    for (int e = 0; e < EPC; ++e) part[(long)blockIdx.y * n + (long)(c0 + t) * EPC + e] = s[e];
  }
}

// The same scatter for many rows (>= 4096: the word and token-type tables of BERT at the bench batch, 78,848 rows each), on SORTED ids
// and without one atomic per element and run: hot ids (token-type ids are all equal; real text is Zipfian) made the kernel above
// serialise on same-address atomics (~800 us per table) and uniformly random ids cost it 60 M atomics.  Here a wave takes EMB_CHUNK
// consecutive rows of the sorted order and sums runs of equal ids; a run that lies strictly between the chunk's first and last
// run belongs to no other chunk and is written with plain stores, while the chunk's first and last run go -- as two (id, partial
// row) entries per chunk -- into a list that is again sorted by id and EMB_CHUNK / 2 times shorter: the kernel runs on that list
// in turn, until the list is short enough for its first / last runs to finish with atomics (78,848 rows: 4,928 entries, then 308).
constexpr int EMB_CHUNK = 32;
constexpr int EMB_FINAL = 1024;   // lists this short end the recursion

template <typename G, bool PERM, bool LAST>
__global__ __launch_bounds__(256) void embedding_bwd_runs_kernel(const G* __restrict__ src, const int64_t* __restrict__ ids,
                                                                 const int64_t* __restrict__ perm, float* __restrict__ dw,
                                                                 float* __restrict__ part_out, int64_t* __restrict__ ids_out,
                                                                 long rows, int d, long vocab) {
  const long w = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long r0 = w * EMB_CHUNK;
  if (r0 >= rows) return;
  const int n = (int)(rows - r0 < EMB_CHUNK ? rows - r0 : EMB_CHUNK);
  const int lane = threadIdx.x & 63;
  const long first_id = ids[r0], last_id = ids[r0 + n - 1];
  if (!LAST && lane == 0) {
    ids_out[2 * w] = first_id;
    ids_out[2 * w + 1] = last_id;
  }
  for (int c0 = lane * 4; c0 < d; c0 += 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    long cur = first_id;
    // a finished run: the chunk's first run -> list entry 2w (or atomics at the last level); any later run that is finished inside
    // the loop lies strictly inside the chunk -> plain store
    auto flush_inner = [&](long id, const float4& a) {
      if (id == first_id) {
        if (LAST) {
          if (id >= 0 && id < vocab) {
            float* o = dw + id * d + c0;
            unsafeAtomicAdd(o, a.x); unsafeAtomicAdd(o + 1, a.y); unsafeAtomicAdd(o + 2, a.z); unsafeAtomicAdd(o + 3, a.w);
          }
        } else {
          *reinterpret_cast<float4*>(part_out + (2 * w) * d + c0) = a;
        }
      } else if (id >= 0 && id < vocab) {
        *reinterpret_cast<float4*>(dw + id * d + c0) = a;
      }
    };
    for (int q0 = 0; q0 < n; q0 += 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (q0 + j < n) {
          const long row = PERM ? perm[r0 + q0 + j] : r0 + q0 + j;
          v[j] = Vec4<G>::load(src + row * d + c0);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (q0 + j < n) {
          const long id = ids[r0 + q0 + j];
          if (id != cur) {
            flush_inner(cur, acc);
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
            cur = id;
          }
          acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w;
        }
      }
    }
    // the chunk's last run (== its first run when the whole chunk is one run: the second entry is then a row of zeros)
    if (cur == first_id) {
      flush_inner(cur, acc);
      if (!LAST) *reinterpret_cast<float4*>(part_out + (2 * w + 1) * d + c0) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else if (LAST) {
      if (cur >= 0 && cur < vocab) {
        float* o = dw + cur * d + c0;
        unsafeAtomicAdd(o, acc.x); unsafeAtomicAdd(o + 1, acc.y); unsafeAtomicAdd(o + 2, acc.z); unsafeAtomicAdd(o + 3, acc.w);
      }
    } else {
      *reinterpret_cast<float4*>(part_out + (2 * w + 1) * d + c0) = acc;
    }
  }
}

static inline long emb_list_len(long rows) { return 2 * ((rows + EMB_CHUNK - 1) / EMB_CHUNK); }

}  // namespace mmk

using namespace mmk;

extern "C" {

int mmk_layernorm_part_blocks(long rows) { return (int)((rows + 4 * LN_ROWS_PER_WAVE - 1) / (4 * LN_ROWS_PER_WAVE)); }

int mmk_layernorm_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int64_t rows, int d,
                      float eps, int dtype, void* stream) {
  // dtype packs (x dtype) | (y dtype << 4)
  MMK_REQUIRE(x && y && mean && rstd && rows >= 0 && d > 0, "bad arguments");
  MMK_REQUIRE(d % 4 == 0 && d <= 64 * 4 * LN_MAX_VEC, "layernorm: d must be a multiple of 4 and <= 2048");
  if (rows == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_LAYERNORM_FWD, st);
  int rc = MMK_DISPATCH_DTYPE(dtype & 15, X, [&]() -> int {
    return MMK_DISPATCH_DTYPE((dtype >> 4) & 15, Y, [&]() -> int {
      if (d <= 128)
        hipLaunchKernelGGL((layernorm_fwd_narrow_kernel<X, Y>), dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, st,
                           static_cast<const X*>(x), w, b, static_cast<Y*>(y), mean, rstd, (long)rows, d, eps);
      else if (d <= 1024)
        hipLaunchKernelGGL((layernorm_fwd_kernel<X, Y, 4>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st,
                           static_cast<const X*>(x), w, b, static_cast<Y*>(y), mean, rstd, (long)rows, d, eps);
      else
        hipLaunchKernelGGL((layernorm_fwd_kernel<X, Y, 8>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st,
                           static_cast<const X*>(x), w, b, static_cast<Y*>(y), mean, rstd, (long)rows, d, eps);
      return 0;
    });
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_layernorm_bwd(const void* x, const void* dy, const float* w, const float* mean, const float* rstd, void* dx, float* part,
                      float* part2, float* dw, float* db, int64_t rows, int d, int dtype, void* stream) {
  // dtype packs (x / dx dtype) | (dy dtype << 4).  part: float[n_blocks, 2, d]; part2: float[256, 2, d] (second stage)
  MMK_REQUIRE(x && dy && mean && rstd && dx && rows >= 0 && d > 0, "bad arguments");
  MMK_REQUIRE(d % 4 == 0 && d <= 64 * 4 * LN_MAX_VEC, "layernorm: d must be a multiple of 4 and <= 2048");
  MMK_REQUIRE((dw == nullptr && db == nullptr) || (part && part2 && dw && db), "dgamma/dbeta need both outputs and the workspaces");
  if (rows == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_LAYERNORM_BWD, st);
  const int n_blocks = mmk_layernorm_part_blocks(rows);
  int rc = MMK_DISPATCH_DTYPE(dtype & 15, X, [&]() -> int {
    return MMK_DISPATCH_DTYPE((dtype >> 4) & 15, G, [&]() -> int {
      if (d <= 128)
        hipLaunchKernelGGL((layernorm_bwd_narrow_kernel<X, G>), dim3(n_blocks), dim3(256), dw ? 16 * d * sizeof(float) : 0, st,
                           static_cast<const X*>(x), static_cast<const G*>(dy), w, mean, rstd, static_cast<X*>(dx),
                           dw ? part : nullptr, (long)rows, d);
      else if (d <= 1024)
        hipLaunchKernelGGL((layernorm_bwd_kernel<X, G, 4>), dim3(n_blocks), dim3(256), dw ? 8 * d * sizeof(float) : 0, st,
                           static_cast<const X*>(x), static_cast<const G*>(dy), w, mean, rstd, static_cast<X*>(dx),
                           dw ? part : nullptr, (long)rows, d);
      else
        hipLaunchKernelGGL((layernorm_bwd_kernel<X, G, 8>), dim3(n_blocks), dim3(256), dw ? 8 * d * sizeof(float) : 0, st,
                           static_cast<const X*>(x), static_cast<const G*>(dy), w, mean, rstd, static_cast<X*>(dx),
                           dw ? part : nullptr, (long)rows, d);
      return 0;
    });
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  if (dw) {
    // part is [n_blocks][2*d]: stage 1 -> up to 256 row slices, stage 2 -> 1
    const int rpb = (n_blocks + COLSUM_SLICES - 1) / COLSUM_SLICES;
    const int slices = (n_blocks + rpb - 1) / rpb;
    hipLaunchKernelGGL(colsum_kernel, dim3((2 * d + 1023) / 1024, slices), dim3(256), 0, st, part, n_blocks, 2 * d, part2, rpb);
    ColsumOuts outs = {{dw, db, nullptr}};
    hipLaunchKernelGGL(colsum_final_kernel, dim3((2 * d + 63) / 64), dim3(256), 0, st, part2, slices, d, 2, outs);
    MMK_LAUNCH_CHECK();
  }
  return 0;
}

#define MMK_ADDLN_FWD(VEC, DROP)                                                                                        \
  hipLaunchKernelGGL((add_layernorm_fwd_kernel<XT, Y, VEC, DROP>), grid, dim3(256), 0, st, static_cast<const XT*>(x), xbias, r, \
                     w, b, s, static_cast<Y*>(y), static_cast<bf16_t*>(y_twin), mean, rstd, (long)rows, d, eps, lo, hi, thr, scale)
#define MMK_ADDLN_BWD(VEC, DROP, XB)                                                                                   \
  hipLaunchKernelGGL((add_layernorm_bwd_kernel<G, XT, VEC, DROP, XB>), dim3(n_blocks), dim3(256), lds, st, s,          \
                     static_cast<const G*>(dy), static_cast<const bf16_t*>(dy_twin), ds_in, w, mean, rstd, dr,           \
                     static_cast<XT*>(dx), dw ? part : nullptr,                                                         \
                     (long)rows, d, lo, hi, thr, scale)

int mmk_add_layernorm_fwd(const void* x, const float* xbias, const float* r, const float* w, const float* b, float* s, void* y,
                          void* y_twin, float* mean, float* rstd, int64_t rows, int d, float eps, int dtype, float dropout_p,
                          uint64_t seed, void* stream) {
  // dtype packs (x dtype) | (y dtype << 4); r and s are f32
  MMK_REQUIRE(x && r && y && mean && rstd && rows >= 0 && d > 0, "bad arguments");
  MMK_REQUIRE(d % 4 == 0 && d <= 64 * 4 * LN_MAX_VEC, "add_layernorm: d must be a multiple of 4 and <= 2048");
  MMK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0, 1)");
  if (rows == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint32_t lo, hi, thr;
  float scale;
  const bool drop = drop_params(dropout_p, seed, &lo, &hi, &thr, &scale);
  ProfScope ps(MMK_K_LAYERNORM_FWD, st);
  const dim3 grid((unsigned)((rows + 3) / 4));
  int rc = MMK_DISPATCH_DTYPE(dtype & 15, XT, [&]() -> int {
    return MMK_DISPATCH_DTYPE((dtype >> 4) & 15, Y, [&]() -> int {
      if (d <= 768) {  // ViT-B / BERT-base width: three float4 per lane, a quarter fewer registers than VEC = 4
        if (drop) MMK_ADDLN_FWD(3, true); else MMK_ADDLN_FWD(3, false);
      } else if (d <= 1024) {
        if (drop) MMK_ADDLN_FWD(4, true); else MMK_ADDLN_FWD(4, false);
      } else {
        if (drop) MMK_ADDLN_FWD(8, true); else MMK_ADDLN_FWD(8, false);
      }
      return 0;
    });
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_add_layernorm_bwd(const float* s, const void* dy, const void* dy_twin, const float* ds_in, const float* w, const float* mean,
                          const float* rstd, float* dr, void* dx, float* part, float* part2, float* dw, float* db, float* dxbias,
                          int64_t rows, int d, int dtype, float dropout_p, uint64_t seed, void* stream) {
  // dtype packs (dx dtype) | (dy dtype << 4).  part: float[n_blocks, NP, d]; part2: float[256, NP, d] (second stage),
  // NP = 3 when dxbias (the column sums of dx = gradient of the deferred Linear bias) is requested, else 2
  MMK_REQUIRE(s && (dy || dy_twin) && mean && rstd && dr && dx && rows >= 0 && d > 0, "bad arguments");
  MMK_REQUIRE(dxbias == nullptr || dw != nullptr, "dxbias is produced together with dgamma/dbeta");
  MMK_REQUIRE(d % 4 == 0 && d <= 64 * 4 * LN_MAX_VEC, "add_layernorm: d must be a multiple of 4 and <= 2048");
  MMK_REQUIRE((dw == nullptr && db == nullptr) || (part && part2 && dw && db), "dgamma/dbeta need both outputs and the workspaces");
  MMK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0, 1)");
  if (rows == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint32_t lo, hi, thr;
  float scale;
  const bool drop = drop_params(dropout_p, seed, &lo, &hi, &thr, &scale);
  ProfScope ps(MMK_K_LAYERNORM_BWD, st);
  const int n_blocks = mmk_layernorm_part_blocks(rows);
  const int np = dxbias ? 3 : 2;
  const size_t lds = dw ? 4 * np * d * sizeof(float) : 0;
  int rc = MMK_DISPATCH_DTYPE(dtype & 15, XT, [&]() -> int {
    return MMK_DISPATCH_DTYPE((dtype >> 4) & 15, G, [&]() -> int {
      if (d <= 768) {
        if (dxbias) { if (drop) MMK_ADDLN_BWD(3, true, true); else MMK_ADDLN_BWD(3, false, true); }
        else { if (drop) MMK_ADDLN_BWD(3, true, false); else MMK_ADDLN_BWD(3, false, false); }
      } else if (d <= 1024) {
        if (dxbias) { if (drop) MMK_ADDLN_BWD(4, true, true); else MMK_ADDLN_BWD(4, false, true); }
        else { if (drop) MMK_ADDLN_BWD(4, true, false); else MMK_ADDLN_BWD(4, false, false); }
      } else {
        if (dxbias) { if (drop) MMK_ADDLN_BWD(8, true, true); else MMK_ADDLN_BWD(8, false, true); }
        else { if (drop) MMK_ADDLN_BWD(8, true, false); else MMK_ADDLN_BWD(8, false, false); }
      }
      return 0;
    });
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  if (dw) {
    const int rpb = (n_blocks + COLSUM_SLICES - 1) / COLSUM_SLICES;
    const int slices = (n_blocks + rpb - 1) / rpb;
    hipLaunchKernelGGL(colsum_kernel, dim3((np * d + 1023) / 1024, slices), dim3(256), 0, st, part, n_blocks, np * d, part2, rpb);
    ColsumOuts outs = {{dw, db, dxbias}};
    hipLaunchKernelGGL(colsum_final_kernel, dim3((np * d + 63) / 64), dim3(256), 0, st, part2, slices, d, np, outs);
    MMK_LAUNCH_CHECK();
  }
  return 0;
}

// column sums of a f32 [n_rows, d] buffer of partial rows, in the two fixed-order stages the row kernels' dgamma / dbeta use
int mmk_colsum_f32(const float* part, int n_rows, int d, float* part2, float* out, void* stream) {
  MMK_REQUIRE(part && part2 && out && n_rows > 0 && d > 0 && d % 4 == 0, "colsum: d must be a multiple of 4");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rpb = (n_rows + COLSUM_SLICES - 1) / COLSUM_SLICES;
  const int slices = (n_rows + rpb - 1) / rpb;
  hipLaunchKernelGGL(colsum_kernel, dim3((d + 1023) / 1024, slices), dim3(256), 0, st, part, n_rows, d, part2, rpb);
  ColsumOuts outs = {{out, nullptr, nullptr}};
  hipLaunchKernelGGL(colsum_final_kernel, dim3((d + 63) / 64), dim3(256), 0, st, part2, slices, d, 1, outs);
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_bias_act_part_blocks(long rows) { return (int)((rows + BA_ROWS - 1) / BA_ROWS); }

int mmk_bias_act_fwd(const void* x, const float* bias, void* y, int64_t rows, int d, int act, int dtype, void* stream) {
  MMK_REQUIRE(x && bias && y && rows >= 0 && d > 0 && d % 8 == 0, "bias_act: d must be a multiple of 8");
  MMK_REQUIRE(act == 0 || act == 1, "bias_act: act must be 0 (quick_gelu) or 1 (gelu)");
  if (rows == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_ACT, st);
  const long n8 = rows * (d / 8);
  const unsigned grid = (unsigned)std::min<long>((n8 + 255) / 256, 256 * 16);
  int rc = MMK_DISPATCH_DTYPE(dtype, T, [&]() -> int {
    if (act == 0)
      hipLaunchKernelGGL((bias_act_fwd_kernel<T, 0>), dim3(grid), dim3(256), 0, st, static_cast<const T*>(x), bias, static_cast<T*>(y), (long)rows, d);
    else
      hipLaunchKernelGGL((bias_act_fwd_kernel<T, 1>), dim3(grid), dim3(256), 0, st, static_cast<const T*>(x), bias, static_cast<T*>(y), (long)rows, d);
    return 0;
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_bias_act_bwd(const void* x, const float* bias, const void* dy, void* dx, float* part, float* part2, float* dbias, int64_t rows,
                     int d, int act, int dtype, void* stream) {
  // part: float[mmk_bias_act_part_blocks(rows), d]; part2: float[256, d]
  MMK_REQUIRE(x && bias && dy && dx && part && part2 && dbias && rows >= 0 && d > 0 && d % 8 == 0, "bias_act: d must be a multiple of 8");
  MMK_REQUIRE(act == 0 || act == 1, "bias_act: act must be 0 (quick_gelu) or 1 (gelu)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (rows == 0) {
    MMK_HIP(hipMemsetAsync(dbias, 0, sizeof(float) * d, st));
    return 0;
  }
  ProfScope ps(MMK_K_ACT, st);
  const int n_blocks = mmk_bias_act_part_blocks(rows);
  int rc = MMK_DISPATCH_DTYPE(dtype, T, [&]() -> int {
    if (act == 0)
      hipLaunchKernelGGL((bias_act_bwd_kernel<T, 0>), dim3(n_blocks), dim3(256), 0, st, static_cast<const T*>(x), bias,
                         static_cast<const T*>(dy), static_cast<T*>(dx), part, (long)rows, d);
    else
      hipLaunchKernelGGL((bias_act_bwd_kernel<T, 1>), dim3(n_blocks), dim3(256), 0, st, static_cast<const T*>(x), bias,
                         static_cast<const T*>(dy), static_cast<T*>(dx), part, (long)rows, d);
    return 0;
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  const int rpb = (n_blocks + COLSUM_SLICES - 1) / COLSUM_SLICES;
  const int slices = (n_blocks + rpb - 1) / rpb;
  hipLaunchKernelGGL(colsum_kernel, dim3((d + 1023) / 1024, slices), dim3(256), 0, st, part, n_blocks, d, part2, rpb);
  ColsumOuts outs = {{dbias, nullptr, nullptr}};
  hipLaunchKernelGGL(colsum_final_kernel, dim3((d + 63) / 64), dim3(256), 0, st, part2, slices, d, 1, outs);
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_quick_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream) {
  MMK_REQUIRE(x && y && n >= 0 && n % 4 == 0, "quick_gelu: n must be a multiple of 4");
  if (n == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_ACT, st);
  const long n4 = n / 4;
  const unsigned grid = (unsigned)std::min<long>((n4 + 255) / 256, 256 * 16);
  int rc = MMK_DISPATCH_DTYPE(dtype, T, [&]() -> int {
    hipLaunchKernelGGL((quick_gelu_fwd_kernel<T>), dim3(grid), dim3(256), 0, st, static_cast<const T*>(x), static_cast<T*>(y), n4);
    return 0;
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_quick_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream) {
  MMK_REQUIRE(x && dy && dx && n >= 0 && n % 4 == 0, "quick_gelu: n must be a multiple of 4");
  if (n == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_ACT, st);
  const long n4 = n / 4;
  const unsigned grid = (unsigned)std::min<long>((n4 + 255) / 256, 256 * 16);
  int rc = MMK_DISPATCH_DTYPE(dtype, T, [&]() -> int {
    hipLaunchKernelGGL((quick_gelu_bwd_kernel<T>), dim3(grid), dim3(256), 0, st, static_cast<const T*>(x), static_cast<const T*>(dy),
                       static_cast<T*>(dx), n4);
    return 0;
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_patchify(const void* in, void* out, int B, int C, int H, int W, int P, int dtype, void* stream) {
  MMK_REQUIRE(in && out && B > 0 && C > 0 && H > 0 && W > 0 && P > 0, "bad arguments");
  MMK_REQUIRE(P % 4 == 0 && H % P == 0 && W % P == 0, "patchify: patch size must be a multiple of 4 and divide H and W");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long n4 = (long)B * (H / P) * (W / P) * (C * P * P / 4);
  const unsigned grid = (unsigned)std::min<long>((n4 + 255) / 256, 256 * 32);
  int rc = MMK_DISPATCH_DTYPE(dtype, T, [&]() -> int {
    hipLaunchKernelGGL((patchify_kernel<T>), dim3(grid), dim3(256), 0, st, static_cast<const T*>(in), static_cast<bf16_t*>(out), B, C, H, W, P);
    return 0;
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_unpatchify(const void* dcols, void* din, int B, int C, int H, int W, int P, int dtype, void* stream) {
  MMK_REQUIRE(dcols && din && B > 0 && C > 0 && H > 0 && W > 0 && P > 0, "bad arguments");
  MMK_REQUIRE(P % 4 == 0 && H % P == 0 && W % P == 0, "unpatchify: patch size must be a multiple of 4 and divide H and W");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long n4 = (long)B * (H / P) * (W / P) * (C * P * P / 4);
  const unsigned grid = (unsigned)std::min<long>((n4 + 255) / 256, 256 * 32);
  const int gd = dtype & 15, td = dtype >> 4;   // gradient-columns dtype | image dtype << 4
  MMK_REQUIRE(gd == MMK_BF16 || gd == MMK_F32, "unpatchify: gradient columns must be bf16 or f32");
  int rc = MMK_DISPATCH_DTYPE(td, T, [&]() -> int {
    if (gd == MMK_BF16)
      hipLaunchKernelGGL((unpatchify_kernel<bf16_t, T>), dim3(grid), dim3(256), 0, st, static_cast<const bf16_t*>(dcols), static_cast<T*>(din), B, C, H, W, P);
    else
      hipLaunchKernelGGL((unpatchify_kernel<float, T>), dim3(grid), dim3(256), 0, st, static_cast<const float*>(dcols), static_cast<T*>(din), B, C, H, W, P);
    return 0;
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_cast_transpose(const void* w, void* w16, void* w16t, int n, int k, int dtype, void* stream) {
  MMK_REQUIRE(w && w16t && n > 0 && k > 0, "bad arguments");
  MMK_REQUIRE(n % 4 == 0 && k % 4 == 0, "cast_transpose: both dimensions must be multiples of 4");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)cdiv(k, CT_TILE), (unsigned)cdiv(n, CT_TILE));
  int rc = MMK_DISPATCH_DTYPE(dtype, T, [&]() -> int {
    hipLaunchKernelGGL((cast_transpose_kernel<T>), grid, dim3(256), 0, st, static_cast<const T*>(w), static_cast<bf16_t*>(w16),
                       static_cast<bf16_t*>(w16t), n, k);
    return 0;
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_embedding_bwd(const void* dout, const int64_t* ids, float* dw, int64_t rows, int d, int64_t vocab, int dtype, void* stream) {
  // dw: f32 [vocab, d], zeroed by the caller; ids outside [0, vocab) are ignored
  MMK_REQUIRE(dout && ids && dw && rows >= 0 && d > 0 && d % 4 == 0 && vocab > 0, "embedding_bwd: d must be a multiple of 4");
  if (rows == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)((rows + 4 * EMB_ROWS - 1) / (4 * EMB_ROWS));
  int rc = MMK_DISPATCH_DTYPE(dtype, G, [&]() -> int {
    hipLaunchKernelGGL((embedding_bwd_kernel<G>), dim3(grid), dim3(256), 0, st, static_cast<const G*>(dout), ids, dw, (long)rows, d, (long)vocab);
    return 0;
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

// y [n_img, h_out, w] = bicubic (A = -0.75, align_corners) stretch of x [n_img, h_in, w] along the middle axis; f32, w % 4 == 0.
// `backward` != 0: x is the gradient w.r.t. y ([n_img, h_out, w]) and y receives the gradient w.r.t. x ([n_img, h_in, w]).
int mmk_cubic_resize_rows(const float* x, float* y, int64_t n_img, int h_in, int h_out, int w, int backward, void* stream) {
  MMK_REQUIRE(x && y && n_img > 0 && h_in > 1 && h_out > 1 && w > 0 && w % 4 == 0, "cubic_resize_rows: f32 [n, h, w] with w % 4 == 0 and h > 1");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float scale = (float)(h_in - 1) / (float)(h_out - 1);
  const long total = n_img * (long)(backward ? h_in : h_out) * (w / 4);
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (backward)
    hipLaunchKernelGGL(cubic_rows_bwd_kernel, dim3(grid), dim3(256), 0, st, x, y, (long)n_img, h_in, h_out, w / 4, scale);
  else
    hipLaunchKernelGGL(cubic_rows_fwd_kernel, dim3(grid), dim3(256), 0, st, x, y, (long)n_img, h_in, h_out, w / 4, scale);
  MMK_LAUNCH_CHECK();
  return 0;
}

// slices (= rows of the f32 workspace `part`, each n wide) mmk_colsum_rows uses for `rows` rows
int mmk_colsum_rows_slices(int64_t rows) { return (int)std::max<int64_t>(1, std::min<int64_t>(CSR_MAX_SLICES, rows / 128)); }

// out[n] (f32) = column sums of x [rows, n] (bf16 / f32, contiguous, n a multiple of 8 / 4): the bias gradient dY.sum(0) of an
// nn.Linear (mmlearn/modules/encoders/*: every Linear whose bias is not folded into a neighbouring kernel).  part: f32 [slices, n].
int mmk_colsum_rows(const void* x, int64_t rows, int n, int dtype, float* part, float* out, void* stream) {
  MMK_REQUIRE(x && part && out && rows > 0 && n > 0, "colsum_rows: bad arguments");
  MMK_REQUIRE(dtype == MMK_F32 ? n % 4 == 0 : (dtype == MMK_BF16 && n % 8 == 0), "colsum_rows: bf16 with n % 8 == 0 or f32 with n % 4 == 0");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int slices = mmk_colsum_rows_slices(rows);
  const long rps = (rows + slices - 1) / slices;
  const int cpr = dtype == MMK_F32 ? n / 4 : n / 8;
  const dim3 grid((cpr + 255) / 256, slices);
  if (dtype == MMK_F32)
    hipLaunchKernelGGL((colsum_rows_kernel<float>), grid, dim3(256), 0, st, static_cast<const float*>(x), (long)rows, n, part, rps);
  else
    hipLaunchKernelGGL((colsum_rows_kernel<bf16_t>), grid, dim3(256), 0, st, static_cast<const bf16_t*>(x), (long)rows, n, part, rps);
  ColsumOuts outs = {{out, nullptr, nullptr}};
  hipLaunchKernelGGL(colsum_final_kernel, dim3((n + 63) / 64), dim3(256), 0, st, part, slices, n, 1, outs);
  MMK_LAUNCH_CHECK();
  return 0;
}

// bytes of scratch mmk_embedding_bwd_sorted needs for `rows` rows of width d: the (id, partial row) lists of every level
int64_t mmk_embedding_bwd_scratch_bytes(int64_t rows, int d) {
  int64_t total = 0;
  for (long n = rows; n > EMB_FINAL;) {
    n = emb_list_len(n);
    total += n * (int64_t)d * 4 + n * 8;
    total = (total + 255) & ~(int64_t)255;
  }
  return total + 256;
}

int mmk_embedding_bwd_sorted(const void* dout, const int64_t* ids_sorted, const int64_t* perm, float* dw, void* scratch, int64_t rows, int d,
                             int64_t vocab, int dtype, void* stream) {
  // dw: f32 [vocab, d], zeroed by the caller; ids_sorted non-decreasing, perm[r] = the row of dout that sorted position r came from;
  // ids outside [0, vocab) are ignored
  MMK_REQUIRE(dout && ids_sorted && perm && dw && scratch && rows >= 0 && d > 0 && d % 4 == 0 && vocab > 0, "embedding_bwd_sorted: d must be a multiple of 4");
  if (rows == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto grid_of = [](long n) { return dim3((unsigned)(((n + EMB_CHUNK - 1) / EMB_CHUNK + 3) / 4)); };
  char* sc = static_cast<char*>(scratch);
  const bool one_level = rows <= EMB_FINAL;
  float* part = nullptr;
  int64_t* pid = nullptr;
  long n = rows;
  if (!one_level) {
    const long n1 = emb_list_len(rows);
    part = reinterpret_cast<float*>(sc);
    pid = reinterpret_cast<int64_t*>(sc + n1 * (int64_t)d * 4);
    sc += ((n1 * (int64_t)d * 4 + n1 * 8) + 255) & ~(int64_t)255;
  }
  int rc = MMK_DISPATCH_DTYPE(dtype, G, [&]() -> int {
    if (one_level)
      hipLaunchKernelGGL((embedding_bwd_runs_kernel<G, true, true>), grid_of(rows), dim3(256), 0, st, static_cast<const G*>(dout), ids_sorted, perm, dw,
                         (float*)nullptr, (int64_t*)nullptr, (long)rows, d, (long)vocab);
    else
      hipLaunchKernelGGL((embedding_bwd_runs_kernel<G, true, false>), grid_of(rows), dim3(256), 0, st, static_cast<const G*>(dout), ids_sorted, perm, dw,
                         part, pid, (long)rows, d, (long)vocab);
    return 0;
  });
  if (rc) return rc;
  if (!one_level) {
    n = emb_list_len(rows);
    while (n > EMB_FINAL) {
      const long n2 = emb_list_len(n);
      float* part2 = reinterpret_cast<float*>(sc);
      int64_t* pid2 = reinterpret_cast<int64_t*>(sc + n2 * (int64_t)d * 4);
      sc += ((n2 * (int64_t)d * 4 + n2 * 8) + 255) & ~(int64_t)255;
      hipLaunchKernelGGL((embedding_bwd_runs_kernel<float, false, false>), grid_of(n), dim3(256), 0, st, part, pid, (const int64_t*)nullptr, dw, part2, pid2, n, d,
                         (long)vocab);
      part = part2;
      pid = pid2;
      n = n2;
    }
    hipLaunchKernelGGL((embedding_bwd_runs_kernel<float, false, true>), grid_of(n), dim3(256), 0, st, part, pid, (const int64_t*)nullptr, dw, (float*)nullptr,
                       (int64_t*)nullptr, n, d, (long)vocab);
  }
  MMK_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
