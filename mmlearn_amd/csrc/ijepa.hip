// HBM-bound row kernels of the I-JEPA path and the L2-normalise op, for gfx950.
//
// Every kernel here is "one wave per row of D elements": coalesced 16-byte lanes, wave-shuffle
// reductions, no LDS tiling needed (no reuse across rows).  They replace:
//   apply_masks                  mmlearn/datasets/processors/masking.py:241-287  (bool-mask gather + host sync)
//   F.layer_norm + apply_masks + repeat_interleave_batch + F.smooth_l1_loss   tasks/ijepa.py:232-238,250-261
//   predictor sequence assembly  modules/encoders/vision.py:545-560
//   ExponentialMovingAverage._update_weights   modules/ema.py:132-158
//   F.normalize                  tasks/contrastive_pretraining.py:428-429
#include <algorithm>

#include "common.h"

namespace mmk {

constexpr int ROWS_PER_BLOCK = 4;  // 4 waves of 64

// ------------------------------------------------------------------ L2 normalise
template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, bf16_t* __restrict__ y16,
                                                         float* __restrict__ inv_norm, int rows, int d) {
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const T* in = x + (size_t)row * d;
  T* out = y + (size_t)row * d;
  const bool vec = (d & 3) == 0;
  float ss = 0.f;
  if (vec) {
    for (int c = lane * 4; c < d; c += 256) {
      const float4 v = Vec4<T>::load(in + c);
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
  } else {
    for (int c = lane; c < d; c += 64) {
      const float v = to_f32(in[c]);
      ss += v * v;
    }
  }
  ss = wave_sum(ss);
  const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
  if (vec) {
    for (int c = lane * 4; c < d; c += 256) {
      float4 v = Vec4<T>::load(in + c);
      v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
      Vec4<T>::store(out + c, v);
      if (y16) Vec4<bf16_t>::store(y16 + (size_t)row * d + c, v);   // the rows as the bf16 similarity kernels will round them
    }
  } else {
    for (int c = lane; c < d; c += 64) {
      const float v = to_f32(in[c]) * inv;
      out[c] = from_f32<T>(v);
      if (y16) y16[(size_t)row * d + c] = from_f32<bf16_t>(v);
    }
  }
}

// dx = (dy - y (y.dy)) * inv  with y = x*inv ; rows whose norm was clamped at eps: dx = dy * inv
template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         const float* __restrict__ inv_norm, T* __restrict__ dx, int rows, int d) {
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const T* xi = x + (size_t)row * d;
  const T* gi = dy + (size_t)row * d;
  T* out = dx + (size_t)row * d;
  const float inv = inv_norm[row];
  float dot = 0.f;
  for (int c = lane; c < d; c += 64) dot += to_f32(xi[c]) * to_f32(gi[c]);
  dot = wave_sum(dot);
  const float proj = (inv < 0.99e12f) ? dot * inv * inv : 0.f;
  for (int c = lane; c < d; c += 64) out[c] = from_f32<T>((to_f32(gi[c]) - to_f32(xi[c]) * proj) * inv);
}

// ------------------------------------------------------------------ mask -> index
__global__ __launch_bounds__(256) void mask_to_index_kernel(const int32_t* __restrict__ mask, int b, int n, int keep,
                                                            int32_t* __restrict__ idx, int32_t* __restrict__ bad) {
  const int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= b) return;
  const int32_t* m = mask + (size_t)row * n;
  int base = 0;
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int c = c0 + lane;
    const bool on = c < n && m[c] != 0;
    const unsigned long long bal = __ballot(on);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (on && base + before < keep) idx[(size_t)row * keep + base + before] = c;
    base += __popcll(bal);
  }
  if (lane == 0 && base != keep) *bad = 1;
}

// ------------------------------------------------------------------ gather / scatter of token rows
// out row (m, bi, p)  <-  x row (bi, idx[m, bi or 0, p]); bytes are copied verbatim (dtype-agnostic)
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ x, char* __restrict__ out,
                                                          const int32_t* __restrict__ idx, int b, int n, int row_bytes,
                                                          int n_masks, int idx_b, int keep) {
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long total = (long)n_masks * b * keep;
  if (row >= total) return;
  const int p = row % keep;
  const int bi = (row / keep) % b;
  const int m = row / ((long)keep * b);
  const int src_tok = idx[((size_t)m * idx_b + (idx_b == 1 ? 0 : bi)) * keep + p];
  const char* s = x + ((size_t)bi * n + src_tok) * row_bytes;
  char* o = out + (size_t)row * row_bytes;
  if ((row_bytes & 15) == 0) {
    for (int c = lane * 16; c < row_bytes; c += 1024) *reinterpret_cast<uint4*>(o + c) = *reinterpret_cast<const uint4*>(s + c);
  } else {
    for (int c = lane * 2; c < row_bytes; c += 128) *reinterpret_cast<uint16_t*>(o + c) = *reinterpret_cast<const uint16_t*>(s + c);
  }
}

// dx row (bi, tok) = sum over masks m that keep tok of dout row (m, bi, pos_m(tok)); idx rows are sorted
// ascending (they come from masks), so membership is a binary search; every dx row is written exactly once.
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const T* __restrict__ dout, T* __restrict__ dx,
                                                           const int32_t* __restrict__ idx, int b, int n, int d, int n_masks,
                                                           int idx_b, int keep) {
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long)b * n) return;
  const int tok = row % n, bi = row / n;
  T* o = dx + (size_t)row * d;
  int src[8];
  int n_src = 0;
  for (int m = 0; m < n_masks && n_src < 8; ++m) {
    const int32_t* ix = idx + ((size_t)m * idx_b + (idx_b == 1 ? 0 : bi)) * keep;
    int lo = 0, hi = keep;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (ix[mid] < tok) lo = mid + 1; else hi = mid;
    }
    if (lo < keep && ix[lo] == tok) src[n_src++] = (m * b + bi) * keep + lo;
  }
  if ((d & 3) == 0) {
    for (int c = lane * 4; c < d; c += 256) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < n_src; ++k) {
        const float4 t = Vec4<T>::load(dout + (size_t)src[k] * d + c);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      Vec4<T>::store(o + c, v);
    }
  } else {
    for (int c = lane; c < d; c += 64) {
      float v = 0.f;
      for (int k = 0; k < n_src; ++k) v += to_f32(dout[(size_t)src[k] * d + c]);
      o[c] = from_f32<T>(v);
    }
  }
}

// ------------------------------------------------------------------ fused target + regression loss
// one wave per (m, bi, p) row: t = LN(h[bi, idx]) in registers/LDS, rho(z - t) reduced.
// MODE 0: forward (block partial sums, optional target store); MODE 1: backward (dz).
template <typename Z, typename H, int MODE>
__global__ __launch_bounds__(256) void ijepa_loss_kernel(const Z* __restrict__ z, const H* __restrict__ h,
                                                         const int32_t* __restrict__ idx, int b, int n, int d, int n_masks,
                                                         int idx_b, int keep, int kind, float eps, H* __restrict__ target_out,
                                                         float* __restrict__ part, const float* __restrict__ upstream,
                                                         Z* __restrict__ dz) {
  extern __shared__ __attribute__((aligned(16))) float rowbuf[];  // [4][d]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave;
  const long total = (long)n_masks * b * keep;
  float local = 0.f;
  if (row < total) {
    const int p = row % keep;
    const int bi = (row / keep) % b;
    const int m = row / ((long)keep * b);
    const int tok = idx[((size_t)m * idx_b + (idx_b == 1 ? 0 : bi)) * keep + p];
    const H* hr = h + ((size_t)bi * n + tok) * d;
    const Z* zr = z + (size_t)row * d;
    float* buf = rowbuf + (size_t)wave * d;
    const float gscale = (MODE == 1) ? (*upstream) / ((float)total * (float)d) : 0.f;
    auto rho = [&](float zv, float t, float& acc, float& g) {
      const float diff = zv - t;
      if (MODE == 0) {
        if (kind == 0) {
          const float ad = fabsf(diff);
          acc += ad < 1.f ? 0.5f * diff * diff : ad - 0.5f;
        } else {
          acc += diff * diff;
        }
      } else {
        g = (kind == 0 ? (fabsf(diff) < 1.f ? diff : (diff > 0.f ? 1.f : -1.f)) : 2.f * diff) * gscale;
      }
    };
    if ((d & 3) == 0) {  // 16-byte (f32) / 8-byte (bf16) lanes
      float s = 0.f;
      for (int c = lane * 4; c < d; c += 256) {
        const float4 v = Vec4<H>::load(hr + c);
        *reinterpret_cast<float4*>(buf + c) = v;
        s += v.x + v.y + v.z + v.w;
      }
      const float mean = wave_sum(s) / d;
      float ss = 0.f;
      for (int c = lane * 4; c < d; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(buf + c);
        const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
        ss += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
      }
      const float rstd = rsqrtf(wave_sum(ss) / d + eps);
      for (int c = lane * 4; c < d; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(buf + c);
        // the reference materialises the target (F.layer_norm output) in h's dtype before the loss
        float4 t = make_float4(to_f32(from_f32<H>((v.x - mean) * rstd)), to_f32(from_f32<H>((v.y - mean) * rstd)),
                               to_f32(from_f32<H>((v.z - mean) * rstd)), to_f32(from_f32<H>((v.w - mean) * rstd)));
        if (MODE == 0 && target_out) Vec4<H>::store(target_out + (size_t)row * d + c, t);
        const float4 zv = Vec4<Z>::load(zr + c);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        rho(zv.x, t.x, local, g.x);
        rho(zv.y, t.y, local, g.y);
        rho(zv.z, t.z, local, g.z);
        rho(zv.w, t.w, local, g.w);
        if (MODE == 1) Vec4<Z>::store(dz + (size_t)row * d + c, g);
      }
    } else {
      float s = 0.f;
      for (int c = lane; c < d; c += 64) {
        const float v = to_f32(hr[c]);
        buf[c] = v;
        s += v;
      }
      const float mean = wave_sum(s) / d;
      float ss = 0.f;
      for (int c = lane; c < d; c += 64) {
        const float v = buf[c] - mean;
        ss += v * v;
      }
      const float rstd = rsqrtf(wave_sum(ss) / d + eps);
      for (int c = lane; c < d; c += 64) {
        const float t = to_f32(from_f32<H>((buf[c] - mean) * rstd));
        if (MODE == 0 && target_out) target_out[(size_t)row * d + c] = from_f32<H>(t);
        float g = 0.f;
        rho(to_f32(zr[c]), t, local, g);
        if (MODE == 1) dz[(size_t)row * d + c] = from_f32<Z>(g);
      }
    }
  }
  if (MODE == 0) {
    __shared__ float red[4];
    local = wave_sum(local);
    if (lane == 0) red[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
  }
}

__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ part, int n, float scale, float* out) {
  float local = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) local += part[i];
  __shared__ float red[4];
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) *out = (red[0] + red[1] + red[2] + red[3]) * scale;
}

// ------------------------------------------------------------------ predictor sequence assembly
// seq row (m, r = e*b + bi, t):  t < n_ctxt : x[r, t] + pos[enc_idx[e, bi, t]]
//                                t >= n_ctxt: mask_token + pos[pred_idx[m, bi, t - n_ctxt]]
template <typename X, typename O>
__global__ __launch_bounds__(256) void pred_assemble_kernel(const X* __restrict__ x, const float* __restrict__ pos,
                                                            const float* __restrict__ tok, const int32_t* __restrict__ enc_idx,
                                                            const int32_t* __restrict__ pred_idx, int b, int d, int n_enc,
                                                            int n_pm, int enc_idx_b, int pred_idx_b, int n_ctxt, int n_pred,
                                                            O* __restrict__ seq) {
  const int L = n_ctxt + n_pred;
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long total = (long)n_pm * n_enc * b * L;
  if (row >= total) return;
  const int t = row % L;
  const long sr = row / L;
  const int r = sr % (n_enc * b);
  const int m = sr / (n_enc * b);
  const int e = r / b, bi = r % b;
  O* o = seq + (size_t)row * d;
  if (t < n_ctxt) {
    const int pt = enc_idx[((size_t)e * enc_idx_b + (enc_idx_b == 1 ? 0 : bi)) * n_ctxt + t];
    const X* xr = x + ((size_t)r * n_ctxt + t) * d;
    const float* pr = pos + (size_t)pt * d;
    // the reference adds in place into x (x's dtype), then torch.cat promotes
    if ((d & 3) == 0) {
      for (int c = lane * 4; c < d; c += 256) {
        const float4 a = Vec4<X>::load(xr + c), q = *reinterpret_cast<const float4*>(pr + c);
        Vec4<O>::store(o + c, make_float4(to_f32(from_f32<X>(a.x + q.x)), to_f32(from_f32<X>(a.y + q.y)),
                                          to_f32(from_f32<X>(a.z + q.z)), to_f32(from_f32<X>(a.w + q.w))));
      }
    } else {
      for (int c = lane; c < d; c += 64) o[c] = from_f32<O>(to_f32(from_f32<X>(to_f32(xr[c]) + pr[c])));
    }
  } else {
    const int pt = pred_idx[((size_t)m * pred_idx_b + (pred_idx_b == 1 ? 0 : bi)) * n_pred + (t - n_ctxt)];
    const float* pr = pos + (size_t)pt * d;
    if ((d & 3) == 0) {
      for (int c = lane * 4; c < d; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(tok + c), q = *reinterpret_cast<const float4*>(pr + c);
        Vec4<O>::store(o + c, make_float4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w));
      }
    } else {
      for (int c = lane; c < d; c += 64) o[c] = from_f32<O>(tok[c] + pr[c]);
    }
  }
}

// dx[r, t] = sum_m dseq[m, r, t] (t < n_ctxt)
template <typename O, typename X>
__global__ __launch_bounds__(256) void pred_assemble_bwd_x_kernel(const O* __restrict__ dseq, int rows_x, int d, int n_pm,
                                                                  int n_ctxt, int n_pred, X* __restrict__ dx) {
  const int L = n_ctxt + n_pred;
  const long row = (long)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long)rows_x * n_ctxt) return;
  const int t = row % n_ctxt;
  const long r = row / n_ctxt;
  if ((d & 3) == 0) {
    for (int c = lane * 4; c < d; c += 256) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int m = 0; m < n_pm; ++m) {
        const float4 q = Vec4<O>::load(dseq + (((size_t)m * rows_x + r) * L + t) * d + c);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      Vec4<X>::store(dx + (size_t)row * d + c, v);
    }
  } else {
    for (int c = lane; c < d; c += 64) {
      float v = 0.f;
      for (int m = 0; m < n_pm; ++m) v += to_f32(dseq[(((size_t)m * rows_x + r) * L + t) * d + c]);
      dx[(size_t)row * d + c] = from_f32<X>(v);
    }
  }
}

// dtok_part[blk, c] = sum over the block's chunk of (sequence, pred-token) rows of dseq[.., n_ctxt + p, c]
template <typename O>
__global__ __launch_bounds__(256) void pred_assemble_bwd_tok_kernel(const O* __restrict__ dseq, long n_seq, int d, int n_ctxt,
                                                                    int n_pred, int rows_per_block, float* __restrict__ part) {
  const int L = n_ctxt + n_pred;
  const long total = n_seq * n_pred;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(total, r0 + rows_per_block);
  for (int c = threadIdx.x; c < d; c += 256) {
    float v = 0.f;
#pragma unroll 8
    for (long q = r0; q < r1; ++q) {
      const long sq = q / n_pred;
      const int p = q % n_pred;
      v += to_f32(dseq[((size_t)sq * L + n_ctxt + p) * d + c]);
    }
    part[(size_t)blockIdx.x * d + c] = v;
  }
}
__global__ __launch_bounds__(256) void sum_cols_kernel(const float* __restrict__ part, int n_blocks, int d, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  float v = 0.f;
#pragma unroll 8
  for (int k = 0; k < n_blocks; ++k) v += part[(size_t)k * d + c];
  out[c] = v;
}

// ------------------------------------------------------------------ multi-tensor EMA
template <typename T>
__device__ __forceinline__ float ld_any(const void* p, long i) {
  return to_f32(static_cast<const T*>(p)[i]);
}
__device__ __forceinline__ float load_tag(const void* p, long i, int dt) {
  return dt == MMK_F32 ? ld_any<float>(p, i) : (dt == MMK_BF16 ? ld_any<bf16_t>(p, i) : ld_any<f16_t>(p, i));
}
__device__ __forceinline__ void store_tag(void* p, long i, int dt, float v) {
  if (dt == MMK_F32) static_cast<float*>(p)[i] = v;
  else if (dt == MMK_BF16) static_cast<bf16_t*>(p)[i] = (bf16_t)v;
  else static_cast<f16_t*>(p)[i] = (f16_t)v;
}
constexpr int EMA_CHUNK = 256 * 16;
// decay_dev != nullptr: the decay is read from a device word (a captured launch then replays with whatever the word holds)
__global__ __launch_bounds__(256) void ema_kernel(const mmk_ema_entry* __restrict__ table, float decay, const float* __restrict__ decay_dev,
                                                  int mode) {
  const mmk_ema_entry e = table[blockIdx.y];
  if (decay_dev != nullptr) decay = decay_dev[0];
  const long base = (long)blockIdx.x * EMA_CHUNK;
  if (base >= e.numel) return;
  const float one_minus = 1.f - decay;
  if (e.teacher_dtype == MMK_F32 && e.student_dtype == MMK_F32 && base + EMA_CHUNK <= e.numel &&
      ((reinterpret_cast<uintptr_t>(e.teacher) | reinterpret_cast<uintptr_t>(e.student)) & 15) == 0) {
    float4* t = reinterpret_cast<float4*>(static_cast<float*>(e.teacher) + base);
    const float4* s = reinterpret_cast<const float4*>(static_cast<const float*>(e.student) + base);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = threadIdx.x + 256 * u;
      float4 sv = s[k];
      if (mode == 1) {
        const float4 tv = t[k];
        // ema.mul_(decay); ema.add_(p, alpha=1-decay)   (modules/ema.py:150-154)
        sv.x = tv.x * decay + sv.x * one_minus;
        sv.y = tv.y * decay + sv.y * one_minus;
        sv.z = tv.z * decay + sv.z * one_minus;
        sv.w = tv.w * decay + sv.w * one_minus;
      }
      t[k] = sv;
    }
    return;
  }
  const long end = min(e.numel, base + EMA_CHUNK);
  for (long i = base + threadIdx.x; i < end; i += 256) {
    float sv = load_tag(e.student, i, e.student_dtype);
    if (mode == 1) sv = load_tag(e.teacher, i, e.teacher_dtype) * decay + sv * one_minus;
    store_tag(e.teacher, i, e.teacher_dtype, sv);
  }
}

// ------------------------------------------------------------------ multi-tensor AdamW (one launch per parameter group)
// torch.optim.AdamW's single-tensor update (torch/optim/adamw.py -> _single_tensor_adam with decoupled weight decay),
// f32 parameters / moments, gradient f32 or bf16:
//   p *= 1 - lr * wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// The foreach implementation is ~7 passes of 15-25 launches over the parameters (3.7 ms for 172 M parameters); this is one
// pass: every block takes one 4096-element chunk from a flat (tensor, offset) work list built once on the host.
// DEV: the learning rate and the step count are read from device words (lr_dev[0], step_dev[0] = the count INCLUDING this step) and
// the bias corrections are formed here, so that a captured launch replays with whatever the words hold at that time
// (torch.optim.AdamW(capturable=True) does the same with its foreach ops).
template <bool DEV>
__global__ __launch_bounds__(256) void adamw_kernel(const mmk_adamw_tensor* __restrict__ tensors, const void* const* __restrict__ grads,
                                                    const int32_t* __restrict__ grad_dtypes, const mmk_adamw_chunk* __restrict__ chunks,
                                                    float lr, float b1, float b2, float eps, float wd, float inv_bc1, float inv_sqrt_bc2,
                                                    const float* __restrict__ lr_dev, const float* __restrict__ step_dev) {
  if (DEV) {
    lr = lr_dev[0];
    const float t = step_dev[0];
    inv_bc1 = 1.f / (1.f - powf(b1, t));
    inv_sqrt_bc2 = 1.f / sqrtf(1.f - powf(b2, t));
  }
  const mmk_adamw_chunk ck = chunks[blockIdx.x];
  const mmk_adamw_tensor t = tensors[ck.tensor];
  const void* g = grads[ck.tensor];
  const int gdt = grad_dtypes[ck.tensor];
  const long base = ck.offset;
  const long end = min(t.numel, base + EMA_CHUNK);
  float* p = static_cast<float*>(t.param);
  float* m = static_cast<float*>(t.exp_avg);
  float* v = static_cast<float*>(t.exp_avg_sq);
  const float decay = 1.f - lr * wd, step = lr * inv_bc1, c1 = 1.f - b1, c2 = 1.f - b2;
  auto upd = [&](float& pv, float& mv, float& vv, float gv) {
    pv *= decay;
    mv = b1 * mv + c1 * gv;
    vv = b2 * vv + c2 * gv * gv;
    pv -= step * mv / (sqrtf(vv) * inv_sqrt_bc2 + eps);
  };
  if (base + EMA_CHUNK <= t.numel && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                                       reinterpret_cast<uintptr_t>(g)) & 15) == 0 && (base & 3) == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long i = base + (threadIdx.x + 256 * u) * 4;
      float4 pv = *reinterpret_cast<float4*>(p + i), mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
      const float4 gv = gdt == MMK_F32 ? *reinterpret_cast<const float4*>(static_cast<const float*>(g) + i)
                                       : Vec4<bf16_t>::load(static_cast<const bf16_t*>(g) + i);
      upd(pv.x, mv.x, vv.x, gv.x); upd(pv.y, mv.y, vv.y, gv.y); upd(pv.z, mv.z, vv.z, gv.z); upd(pv.w, mv.w, vv.w, gv.w);
      *reinterpret_cast<float4*>(p + i) = pv;
      *reinterpret_cast<float4*>(m + i) = mv;
      *reinterpret_cast<float4*>(v + i) = vv;
    }
    return;
  }
  for (long i = base + threadIdx.x; i < end; i += 256) {
    float pv = p[i], mv = m[i], vv = v[i];
    upd(pv, mv, vv, load_tag(g, i, gdt));
    p[i] = pv; m[i] = mv; v[i] = vv;
  }
}

}  // namespace mmk

using namespace mmk;

extern "C" {

int mmk_l2norm_fwd_twin(const void* x, void* y, void* y16, float* inv_norm, int rows, int d, int dtype, void* stream) {
  MMK_REQUIRE(x && y && rows >= 0 && d > 0, "bad arguments");
  if (rows == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_L2NORM, st);
  int rc = MMK_DISPATCH_DTYPE(dtype, T, [&]() -> int {
    hipLaunchKernelGGL((l2norm_fwd_kernel<T>), dim3(cdiv(rows, ROWS_PER_BLOCK)), dim3(256), 0, st, static_cast<const T*>(x),
                       static_cast<T*>(y), static_cast<bf16_t*>(y16), inv_norm, rows, d);
    return 0;
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_l2norm_fwd(const void* x, void* y, float* inv_norm, int rows, int d, int dtype, void* stream) {
  return mmk_l2norm_fwd_twin(x, y, nullptr, inv_norm, rows, d, dtype, stream);
}

int mmk_l2norm_bwd(const void* x, const void* dy, const float* inv_norm, void* dx, int rows, int d, int dtype, void* stream) {
  MMK_REQUIRE(x && dy && inv_norm && dx && rows >= 0 && d > 0, "bad arguments");
  if (rows == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_L2NORM, st);
  int rc = MMK_DISPATCH_DTYPE(dtype, T, [&]() -> int {
    hipLaunchKernelGGL((l2norm_bwd_kernel<T>), dim3(cdiv(rows, ROWS_PER_BLOCK)), dim3(256), 0, st, static_cast<const T*>(x),
                       static_cast<const T*>(dy), inv_norm, static_cast<T*>(dx), rows, d);
    return 0;
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_mask_to_index(const int32_t* mask, int b, int n, int keep, int32_t* idx, int32_t* bad, void* stream) {
  MMK_REQUIRE(mask && idx && bad && b > 0 && n > 0 && keep > 0 && keep <= n, "bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_MASK_INDEX, st);
  hipLaunchKernelGGL(mask_to_index_kernel, dim3(cdiv(b, ROWS_PER_BLOCK)), dim3(256), 0, st, mask, b, n, keep, idx, bad);
  MMK_LAUNCH_CHECK();
  return 0;
}

static int check_gather(int b, int n, int d, int n_masks, int idx_b, int keep) {
  MMK_REQUIRE(b > 0 && n > 0 && d > 0 && n_masks > 0 && keep > 0 && keep <= n, "bad shape");
  MMK_REQUIRE(idx_b == 1 || idx_b == b, "idx batch dim must be 1 or b");
  return 0;
}

int mmk_gather_rows(const void* x, void* out, const int32_t* idx, int b, int n, int d, int n_masks, int idx_b, int keep,
                    int dtype, void* stream) {
  MMK_REQUIRE(x && out && idx, "null pointer");
  int rc = check_gather(b, n, d, n_masks, idx_b, keep);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_GATHER_ROWS, st);
  const long total = (long)n_masks * b * keep;
  const int row_bytes = d * (int)dtype_size(dtype);
  MMK_REQUIRE((row_bytes & 1) == 0, "row bytes must be even");
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), 0, st,
                     static_cast<const char*>(x), static_cast<char*>(out), idx, b, n, row_bytes, n_masks, idx_b, keep);
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_scatter_rows(const void* dout, void* dx, const int32_t* idx, int b, int n, int d, int n_masks, int idx_b, int keep,
                     int dtype, void* stream) {
  MMK_REQUIRE(dout && dx && idx, "null pointer");
  int rc = check_gather(b, n, d, n_masks, idx_b, keep);
  if (rc) return rc;
  MMK_REQUIRE(n_masks <= 8, "at most 8 masks per scatter");
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_SCATTER_ROWS, st);
  const long total = (long)b * n;
  rc = MMK_DISPATCH_DTYPE(dtype, T, [&]() -> int {
    hipLaunchKernelGGL((scatter_rows_kernel<T>), dim3((unsigned)((total + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), 0,
                       st, static_cast<const T*>(dout), static_cast<T*>(dx), idx, b, n, d, n_masks, idx_b, keep);
    return 0;
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_ijepa_loss_blocks(int rows) { return cdiv(rows, ROWS_PER_BLOCK); }

int mmk_ijepa_loss_fwd(const void* z, const void* h, const int32_t* idx, int b, int n, int d, int n_masks, int idx_b, int keep,
                       int dtype, int kind, float eps, void* target_out, float* part, int n_blocks, float* loss, void* stream) {
  MMK_REQUIRE(z && h && idx && part && loss, "null pointer");
  int rc = check_gather(b, n, d, n_masks, idx_b, keep);
  if (rc) return rc;
  MMK_REQUIRE(kind == 0 || kind == 1, "kind must be 0 (smooth-L1) or 1 (MSE)");
  const long total = (long)n_masks * b * keep;
  MMK_REQUIRE(n_blocks == cdiv((int)total, ROWS_PER_BLOCK), "n_blocks mismatch (use mmk_ijepa_loss_blocks)");
  MMK_REQUIRE((size_t)d * 4 * ROWS_PER_BLOCK <= 64 * 1024, "d too large for the row buffer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_IJEPA_LOSS_FWD, st);
  rc = MMK_DISPATCH_DTYPE(dtype & 15, Z, [&]() -> int {
    return MMK_DISPATCH_DTYPE((dtype >> 4) & 15, H, [&]() -> int {
      hipLaunchKernelGGL((ijepa_loss_kernel<Z, H, 0>), dim3(n_blocks), dim3(256), ROWS_PER_BLOCK * d * sizeof(float), st,
                         static_cast<const Z*>(z), static_cast<const H*>(h), idx, b, n, d, n_masks, idx_b, keep, kind, eps,
                         static_cast<H*>(target_out), part, nullptr, nullptr);
      return 0;
    });
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(256), 0, st, part, n_blocks, 1.f / ((float)total * (float)d), loss);
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_ijepa_loss_bwd(const void* z, const void* h, const int32_t* idx, int b, int n, int d, int n_masks, int idx_b, int keep,
                       int dtype, int kind, float eps, const float* upstream, void* dz, void* stream) {
  MMK_REQUIRE(z && h && idx && upstream && dz, "null pointer");
  int rc = check_gather(b, n, d, n_masks, idx_b, keep);
  if (rc) return rc;
  MMK_REQUIRE((size_t)d * 4 * ROWS_PER_BLOCK <= 64 * 1024, "d too large for the row buffer");
  const long total = (long)n_masks * b * keep;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_IJEPA_LOSS_BWD, st);
  rc = MMK_DISPATCH_DTYPE(dtype & 15, Z, [&]() -> int {
    return MMK_DISPATCH_DTYPE((dtype >> 4) & 15, H, [&]() -> int {
      hipLaunchKernelGGL((ijepa_loss_kernel<Z, H, 1>), dim3(cdiv((int)total, ROWS_PER_BLOCK)), dim3(256),
                         ROWS_PER_BLOCK * d * sizeof(float), st, static_cast<const Z*>(z), static_cast<const H*>(h), idx, b, n,
                         d, n_masks, idx_b, keep, kind, eps, nullptr, nullptr, upstream, static_cast<Z*>(dz));
      return 0;
    });
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_pred_tok_blocks(int rows) { return std::min(256, std::max(1, cdiv(rows, 16))); }

int mmk_pred_assemble(const void* x, const void* pos, const void* mask_token, const int32_t* enc_idx, const int32_t* pred_idx,
                      int b, int n, int d, int n_enc, int n_pred_masks, int enc_idx_b, int pred_idx_b, int n_ctxt, int n_pred,
                      int dtype, void* seq, void* stream) {
  // dtype packs (x dtype) | (seq dtype << 4); pos and mask_token are f32
  const int xdt = dtype & 15, odt = (dtype >> 4) & 15;
  MMK_REQUIRE(x && pos && mask_token && enc_idx && pred_idx && seq, "null pointer");
  MMK_REQUIRE(b > 0 && n > 0 && d > 0 && n_enc > 0 && n_pred_masks > 0 && n_ctxt > 0 && n_pred > 0, "bad shape");
  MMK_REQUIRE((enc_idx_b == 1 || enc_idx_b == b) && (pred_idx_b == 1 || pred_idx_b == b), "idx batch dim must be 1 or b");
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_PRED_ASSEMBLE, st);
  const long total = (long)n_pred_masks * n_enc * b * (n_ctxt + n_pred);
  const unsigned grid = (unsigned)((total + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
  int rc = MMK_DISPATCH_DTYPE(xdt, X, [&]() -> int {
    return MMK_DISPATCH_DTYPE(odt, O, [&]() -> int {
      hipLaunchKernelGGL((pred_assemble_kernel<X, O>), dim3(grid), dim3(256), 0, st, static_cast<const X*>(x),
                         static_cast<const float*>(pos), static_cast<const float*>(mask_token), enc_idx, pred_idx, b, d, n_enc,
                         n_pred_masks, enc_idx_b, pred_idx_b, n_ctxt, n_pred, static_cast<O*>(seq));
      return 0;
    });
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_pred_assemble_bwd(const void* dseq, int b, int d, int n_enc, int n_pred_masks, int n_ctxt, int n_pred, int dtype,
                          void* dx, float* dtok_part, int n_tok_blocks, void* dtok, void* stream) {
  const int xdt = dtype & 15, odt = (dtype >> 4) & 15;
  MMK_REQUIRE(dseq && b > 0 && d > 0 && n_enc > 0 && n_pred_masks > 0 && n_ctxt > 0 && n_pred > 0, "bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_PRED_ASSEMBLE_BWD, st);
  const int rows_x = n_enc * b;
  const long n_seq = (long)n_pred_masks * rows_x;
  int rc = 0;
  if (dx) {
    const long total = (long)rows_x * n_ctxt;
    rc = MMK_DISPATCH_DTYPE(odt, O, [&]() -> int {
      return MMK_DISPATCH_DTYPE(xdt, X, [&]() -> int {
        hipLaunchKernelGGL((pred_assemble_bwd_x_kernel<O, X>), dim3((unsigned)((total + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)),
                           dim3(256), 0, st, static_cast<const O*>(dseq), rows_x, d, n_pred_masks, n_ctxt, n_pred,
                           static_cast<X*>(dx));
        return 0;
      });
    });
    if (rc) return rc;
    MMK_LAUNCH_CHECK();
  }
  if (dtok) {
    MMK_REQUIRE(dtok_part && n_tok_blocks > 0, "missing token-gradient workspace");
    const long total = n_seq * n_pred;
    const int rpb = (int)((total + n_tok_blocks - 1) / n_tok_blocks);
    rc = MMK_DISPATCH_DTYPE(odt, O, [&]() -> int {
      hipLaunchKernelGGL((pred_assemble_bwd_tok_kernel<O>), dim3(n_tok_blocks), dim3(256), 0, st, static_cast<const O*>(dseq),
                         n_seq, d, n_ctxt, n_pred, rpb, dtok_part);
      return 0;
    });
    if (rc) return rc;
    MMK_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_cols_kernel, dim3(cdiv(d, 256)), dim3(256), 0, st, dtok_part, n_tok_blocks, d,
                       static_cast<float*>(dtok));
    MMK_LAUNCH_CHECK();
  }
  return 0;
}

int mmk_ema_update(const mmk_ema_entry* table, int n_tensors, int64_t max_numel, float decay, int mode, void* stream) {
  MMK_REQUIRE(table && n_tensors > 0 && max_numel > 0, "bad arguments");
  MMK_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (copy) or 1 (ema)");
  MMK_REQUIRE(n_tensors <= 65535, "too many tensors for one launch");
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_EMA, st);
  const long chunks = (max_numel + EMA_CHUNK - 1) / EMA_CHUNK;
  hipLaunchKernelGGL(ema_kernel, dim3((unsigned)chunks, n_tensors), dim3(256), 0, st, table, decay, static_cast<const float*>(nullptr), mode);
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_ema_update_dev(const mmk_ema_entry* table, int n_tensors, int64_t max_numel, const float* decay_dev, int mode, void* stream) {
  MMK_REQUIRE(table && decay_dev && n_tensors > 0 && max_numel > 0, "bad arguments");
  MMK_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (copy) or 1 (ema)");
  MMK_REQUIRE(n_tensors <= 65535, "too many tensors for one launch");
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_EMA, st);
  const long chunks = (max_numel + EMA_CHUNK - 1) / EMA_CHUNK;
  hipLaunchKernelGGL(ema_kernel, dim3((unsigned)chunks, n_tensors), dim3(256), 0, st, table, 0.f, decay_dev, mode);
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_adamw_chunk_elems(void) { return EMA_CHUNK; }

int mmk_adamw_update(const mmk_adamw_tensor* tensors, const void* const* grads, const int32_t* grad_dtypes, const mmk_adamw_chunk* chunks,
                     int n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream) {
  MMK_REQUIRE(tensors && grads && grad_dtypes && chunks && n_chunks > 0 && step > 0, "bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adamw_kernel<false>, dim3((unsigned)n_chunks), dim3(256), 0, st, tensors, grads, grad_dtypes, chunks, lr, beta1, beta2, eps,
                     weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), static_cast<const float*>(nullptr),
                     static_cast<const float*>(nullptr));
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_adamw_update_dev(const mmk_adamw_tensor* tensors, const void* const* grads, const int32_t* grad_dtypes, const mmk_adamw_chunk* chunks,
                         int n_chunks, const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, const float* step_dev,
                         void* stream) {
  MMK_REQUIRE(tensors && grads && grad_dtypes && chunks && n_chunks > 0 && lr_dev && step_dev, "bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(adamw_kernel<true>, dim3((unsigned)n_chunks), dim3(256), 0, st, tensors, grads, grad_dtypes, chunks, 0.f, beta1, beta2, eps,
                     weight_decay, 0.f, 0.f, lr_dev, step_dev);
  MMK_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
