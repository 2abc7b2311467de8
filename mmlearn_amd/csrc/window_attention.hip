// SURVEY 8(f1) / BASELINE configs[3]: the windowed self-attention of HTSAT (the audio tower mmlearn's three-modality contrastive
// configuration pulls in through HF CLAP; Swin-style: 8 x 8 = 64-token windows, head dim 24, a learned relative-position bias per
// head and an additive mask per window position in the shifted layers):
//
//     S = scale * Q K^T + bias[h] (+ mask[w]),   P = softmax_j(S),   O = P V          per (window, head): [64 x 24] operands
//
// HF runs it as two batched GEMMs of 64 x 64 x 24, an f32 softmax, an add per table and the copies between them -- 38 % of the tower's
// forward + backward at batch 256.  Here one wave owns one (window, head) item at a time: its Q, K, V rows (48-byte segments of the
// [windows * 64, C] projection outputs) arrive by LDS-DMA as a LINEAR image (chunk c of 16 bytes = row c / 3, piece c % 3, lands at
// byte 16 c, which IS the row-major [64][24] tile), S^T = K Q^T is formed with the query on the lane (v_mfma_f32_32x32x8_bf16_1k:
// 3 k-steps cover the head dim exactly), the softmax runs over registers + one exchange with lane ^ 32, and the normalised
// accumulators are the B operand of O^T = V^T P^T with V^T fragments by ds_read_b64_tr_b16.  Four waves of a workgroup share the
// (window position, head) and with it ONE [64][64] f32 table bias[h] + mask[w] in LDS (pre-multiplied by log2 e, 16-byte chunks
// XOR-swizzled by the row so that both the query-on-lane and the key-on-lane reads are conflict-free); they differ in the batch sample.
// The blockIdx -> (window position, batch slice, head) map keeps the heads that share 128-byte lines on one XCD.
//
// Backward (one kernel, nothing but lse kept from the forward -- not even O): S and dP = dO V^T are formed in BOTH orientations --
// query on the lane for dQ^T = K^T dS^T, key on the lane for dV^T = dO^T P and dK^T = Q^T dS -- so that dS / P are always the
// accumulator-as-operand and the other operand a transposed read; delta_i = <dO_i, O_i> = sum_j P_ij dP_ij falls out of the first
// orientation's registers; the bias gradient, sum over items of dS, stays in 64 registers per wave across the items of the
// workgroup and leaves as one [64][64] partial per workgroup.  The pass is HBM-bound by design: 4 reads + 3 writes of [64][24]
// per item, ~100 small MFMAs.
#include <hip/hip_ext.h>
#include <stdint.h>

#include "common.h"

namespace mmk {

typedef float wa_f32x16 __attribute__((ext_vector_type(16)));
typedef short wa_s16x4 __attribute__((ext_vector_type(4)));

constexpr int WA_N = 64;                      // tokens per window
constexpr int WA_TAB = WA_N * WA_N * 4;       // bytes of the f32 table
constexpr float WA_LOG2E = 1.4426950408889634f;

struct WinAttnArgs {
  const bf16_t* q;      // [B * nW, 64, C] contiguous; head hh = columns hh * DH .. + DH
  const bf16_t* k;
  const bf16_t* v;
  bf16_t* o;            // [B * nW, 64, C]
  float* lse2;          // [B * nW, H, 64]: log2 of the softmax denominators (base-2 logits)
  const float* table;   // [nWt, H, 64, 64] f32: bias[h] (+ mask[w]); nWt = 1 (no mask) or nW
  // backward only
  const bf16_t* dout;   // [B * nW, 64, C]
  bf16_t* dq;
  bf16_t* dk;
  bf16_t* dv;
  float* dtab_part;     // [gridDim.x][64][64] f32: per-workgroup sums of dS (block id = the forward's)
  int B, nW, nWt, H, C, nsplit;
  int img_h, img_w, shift;   // img_w > 0: q .. dv are [B, img_h * img_w, C] token maps and the windows are gathered from / scattered to them (below)
  int ld;                    // row stride (elements) of q, k, v, dq, dk, dv: C, or 3 C when they are the thirds of one packed [.., 3 C] projection
  float scale;
};

__device__ __forceinline__ void wa_dma16(const void* sbase, uint32_t voff, uint32_t lds_addr) {
  const uint64_t pb = reinterpret_cast<uint64_t>(sbase);
  const uint64_t ps = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pb >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pb);
  lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(reinterpret_cast<const void*>(ps)), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint32_t wa_lds_addr(const char* p) {
  return (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)p);
}
__device__ __forceinline__ void wa_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// 4 consecutive bf16 of one row (8 bytes): the 32x32x8 A / B operand of lane (r, h) is X[row r][k = 8 ks + 4 h .. + 3]
__device__ __forceinline__ wa_s16x4 wa_row_frag(const char* tile, int rowb, int row, int ks, int h) {
  return *reinterpret_cast<const wa_s16x4*>(tile + row * rowb + (8 * ks + 4 * h) * 2);
}
// transposed fragment: A operand X^T[d = lane & 31][k = 4 h + q] = X[row0 + 4 h + q][d]; `tr_off` = this lane's address inside
// the 4-row x 16-column block its 16-lane group gathers (row q = (lane & 15) >> 2, columns 16 gi + 4 p)
__device__ __forceinline__ wa_s16x4 wa_tr_frag(const char* tile, int rowb, int row0, int h, int tr_off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wa_s16x4*)(tile + (row0 + 4 * h) * rowb + tr_off));
}
__device__ __forceinline__ wa_s16x4 wa_pack4(float a, float b, float c, float d) {
  typedef bf16_t bf4 __attribute__((ext_vector_type(4)));
  bf4 v;
  v[0] = (bf16_t)a; v[1] = (bf16_t)b; v[2] = (bf16_t)c; v[3] = (bf16_t)d;
  return __builtin_bit_cast(wa_s16x4, v);
}
__device__ __forceinline__ wa_f32x16 wa_mfma(wa_s16x4 a, wa_s16x4 b, wa_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
}

// blockIdx -> (pair = window position + nW * batch slice, head).  With the pair count a multiple of 8 the H heads of a pair get block
// ids 8 apart: the hardware deals consecutive ids round-robin to the 8 XCDs, so they land on ONE XCD, next to each other in
// dispatch order, and the 128-byte lines their 48-byte row segments share are fetched into that L2 once.
__device__ __forceinline__ void wa_decode_block(const WinAttnArgs& a, int& pair, int& hh) {
  const int id = blockIdx.x, npairs = a.nW * a.nsplit;
  if ((npairs & 7) == 0) {
    const int x = id & 7, t = id >> 3;
    hh = t % a.H;
    pair = (t / a.H) * 8 + x;
  } else {
    pair = id / a.H;
    hh = id % a.H;
  }
}

// the workgroup's table -> LDS: row i, 16-byte chunk c (keys 4 c .. 4 c + 3) at chunk position c ^ (i & 15), times log2(e)
__device__ __forceinline__ void wa_load_table(const WinAttnArgs& a, float* tab, int w, int hh) {
  const float* tg = a.table + ((long)(a.nWt > 1 ? w : 0) * a.H + hh) * (WA_N * WA_N);
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int ci = threadIdx.x + 256 * n, i = ci >> 4, c = ci & 15;
    float4 v = *reinterpret_cast<const float4*>(tg + i * WA_N + c * 4);
    v.x *= WA_LOG2E; v.y *= WA_LOG2E; v.z *= WA_LOG2E; v.w *= WA_LOG2E;
    *reinterpret_cast<float4*>(tab + i * WA_N + ((c ^ (i & 15)) << 2)) = v;
  }
}

// accumulators with the tile's columns on the lanes and rows d = 8 g + 4 h + t in registers 4 g + t  ->  bf16 rows [col][d] of a
// [64][DH] LDS tile (8-byte pieces), `mul` applied
template <int DH>
__device__ __forceinline__ void wa_stage_out(char* stg, int col, int h, const wa_f32x16& acc, float mul) {
#pragma unroll
  for (int g = 0; g < DH / 8; ++g)
    *reinterpret_cast<wa_s16x4*>(stg + col * (DH * 2) + (8 * g + 4 * h) * 2) =
        wa_pack4(acc[4 * g] * mul, acc[4 * g + 1] * mul, acc[4 * g + 2] * mul, acc[4 * g + 3] * mul);
}
// the staged [64][DH] tile -> global rows (16-byte chunks, the same linear chunk order the loads use)
template <int DH>
__device__ __forceinline__ void wa_store_tile(const char* stg, bf16_t* dst, const uint32_t (&voff)[DH / 8], int lane) {
#pragma unroll
  for (int n = 0; n < DH / 8; ++n) {
    const uint4 val = *reinterpret_cast<const uint4*>(stg + 16 * (lane + 64 * n));
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(dst) + voff[n]) = val;
  }
}

// Byte offsets, from the sample's (token-map mode) or the window's (window mode) first row of this head, of the 16-byte chunks
// c = lane + 64 n of the item's [64][DH] tile: chunk c = token c / CH, piece c % CH.  Window mode: the windows are contiguous
// [64, C] blocks (HF's window_partition output).  Token-map mode: token (ty, tx) of window (wy, wx) of the image rolled by
// -shift is row ((8 wy + ty + shift) mod img_h) * img_w + (8 wx + tx + shift) mod img_w of the sample -- torch.roll, window_partition
// and, on the way out, window_reverse and the roll back (HF ClapAudioLayer.forward) as address arithmetic.  Same for every item
// of the workgroup (they differ in the sample only).
template <int CH>
__device__ __forceinline__ void wa_chunk_offsets(const WinAttnArgs& a, int w, int lane, int ld, uint32_t (&voff)[CH]) {
#pragma unroll
  for (int n = 0; n < CH; ++n) {
    const int c = lane + 64 * n, t = c / CH, piece = c % CH;
    int row = t;
    if (a.img_w > 0) {
      const int wpr = a.img_w >> 3, wy = w / wpr, wx = w % wpr;
      const int y = (8 * wy + (t >> 3) + a.shift) % a.img_h, x = (8 * wx + (t & 7) + a.shift) % a.img_w;
      row = y * a.img_w + x;
    }
    voff[n] = (uint32_t)(row * ld + piece * 8) * 2u;
  }
}

template <int DH>
__global__ __launch_bounds__(256, 3) void win_attn_fwd_kernel(const WinAttnArgs a) {
  constexpr int ROWB = DH * 2, CH = DH / 8, KS = DH / 8, TILE = WA_N * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tab = reinterpret_cast<float*>(smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  char* Qs = smem + WA_TAB + wave * (3 * TILE + 64);
  char* Ks = Qs + TILE;
  char* Vs = Ks + TILE;
  int pair, hh;
  wa_decode_block(a, pair, hh);
  const int w = pair % a.nW, slice = pair / a.nW;
  wa_load_table(a, tab, w, hh);
  __syncthreads();
  const int r = lane & 31, h = lane >> 5;
  const int tr_off = ((lane & 15) >> 2) * ROWB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  uint32_t voff[CH], voff_o[CH];   // chunk offsets in the q / k / v tensors (row stride ld) and in o (row stride C)
  wa_chunk_offsets<CH>(a, w, lane, a.ld, voff);
  wa_chunk_offsets<CH>(a, w, lane, a.C, voff_o);
  const float sc2 = a.scale * WA_LOG2E;
  const int bps = (a.B + a.nsplit - 1) / a.nsplit;
  const int b0 = slice * bps, b1 = min(a.B, b0 + bps);
  for (int b = b0 + wave; b < b1; b += 4) {
    const long bw = (long)b * a.nW + w;
    const long rbase = (a.img_w > 0 ? (long)b * a.nW : bw) * WA_N;   // first row of the item's window (window mode) or sample (token-map mode)
    const long ebase = rbase * a.ld + hh * DH, obase = rbase * a.C + hh * DH;
#pragma unroll
    for (int n = 0; n < CH; ++n) {
      wa_dma16(a.q + ebase, voff[n], wa_lds_addr(Qs) + n * 1024);
      wa_dma16(a.k + ebase, voff[n], wa_lds_addr(Ks) + n * 1024);
      wa_dma16(a.v + ebase, voff[n], wa_lds_addr(Vs) + n * 1024);
    }
    wa_wait_dma();
    // ---- S^T tiles (key tile jt, query tile it): rows = keys in registers, column = query i = 32 it + r on the lane
    wa_f32x16 s[2][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        wa_f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          acc = wa_mfma(wa_row_frag(Ks, ROWB, 32 * jt + r, ks, h), wa_row_frag(Qs, ROWB, 32 * it + r, ks, h), acc);
        s[jt][it] = acc;
      }
    wa_f32x16 oacc[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = 32 * it + r;
      float m = -INFINITY;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 t = *reinterpret_cast<const float4*>(tab + i * WA_N + (((8 * jt + 2 * g + h) ^ (i & 15)) << 2));
          s[jt][it][4 * g] = fmaf(s[jt][it][4 * g], sc2, t.x);
          s[jt][it][4 * g + 1] = fmaf(s[jt][it][4 * g + 1], sc2, t.y);
          s[jt][it][4 * g + 2] = fmaf(s[jt][it][4 * g + 2], sc2, t.z);
          s[jt][it][4 * g + 3] = fmaf(s[jt][it][4 * g + 3], sc2, t.w);
          m = fmaxf(m, fmaxf(fmaxf(s[jt][it][4 * g], s[jt][it][4 * g + 1]), fmaxf(s[jt][it][4 * g + 2], s[jt][it][4 * g + 3])));
        }
      m = fmaxf(m, __shfl_xor(m, 32));
      float sum = 0.f;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float p = __builtin_amdgcn_exp2f(s[jt][it][e] - m);
          s[jt][it][e] = p;
          sum += p;
        }
      sum += __shfl_xor(sum, 32);
      const float inv = 1.f / sum;
      if (h == 0) a.lse2[(bw * a.H + hh) * WA_N + i] = m + __builtin_amdgcn_logf(sum);   // v_log_f32 = log2
      // ---- O^T[d][i] = sum_j V^T[d][j] P^T[j][i]
      wa_f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc = wa_mfma(wa_tr_frag(Vs, ROWB, 32 * jt + 8 * g, h, tr_off),
                        wa_pack4(s[jt][it][4 * g] * inv, s[jt][it][4 * g + 1] * inv, s[jt][it][4 * g + 2] * inv, s[jt][it][4 * g + 3] * inv), acc);
      oacc[it] = acc;
    }
    // ---- rows of O through the (now free) Q tile
#pragma unroll
    for (int it = 0; it < 2; ++it) wa_stage_out<DH>(Qs, 32 * it + r, h, oacc[it], 1.f);
    wa_store_tile<DH>(Qs, a.o + obase, voff_o, lane);
  }
}

template <int DH>
__global__ __launch_bounds__(256, 2) void win_attn_bwd_kernel(const WinAttnArgs a) {
  constexpr int ROWB = DH * 2, CH = DH / 8, KS = DH / 8, TILE = WA_N * ROWB;
  constexpr int WAVE_LDS = 5 * TILE + 64 + 512;   // Q, K, V, dO tiles, slack for the transposed reads, a staging tile, lse2[64] and delta[64]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tab = reinterpret_cast<float*>(smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  char* Qs = smem + WA_TAB + wave * WAVE_LDS;
  char* Ks = Qs + TILE;
  char* Vs = Ks + TILE;
  char* Gs = Vs + TILE;        // dO
  char* Os = Gs + TILE + 64;   // staging tile of dQ
  float* lses = reinterpret_cast<float*>(Os + TILE);
  float* dels = lses + WA_N;
  int pair, hh;
  wa_decode_block(a, pair, hh);
  const int w = pair % a.nW, slice = pair / a.nW;
  wa_load_table(a, tab, w, hh);
  __syncthreads();
  const int h = lane >> 5;
  const int tr_off = ((lane & 15) >> 2) * ROWB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  uint32_t voff[CH], voff_o[CH];   // chunk offsets in q / k / v / dq / dk / dv (row stride ld) and in dO (row stride C)
  wa_chunk_offsets<CH>(a, w, lane, a.ld, voff);
  wa_chunk_offsets<CH>(a, w, lane, a.C, voff_o);
  const float sc2 = a.scale * WA_LOG2E;
  const int bps = (a.B + a.nsplit - 1) / a.nsplit;
  const int b0 = slice * bps, b1 = min(a.B, b0 + bps);
  // sum over this wave's items of dS[i][j], key j = 32 jt + r on the lane, query i = 32 it + 8 g + 4 h + t in register 4 g + t of dT[it][jt]
  wa_f32x16 dT[2][2];
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) dT[it][jt][e] = 0.f;
  for (int b = b0 + wave; b < b1; b += 4) {
    const long bw = (long)b * a.nW + w;
    const long rbase = (a.img_w > 0 ? (long)b * a.nW : bw) * WA_N;
    const long ebase = rbase * a.ld + hh * DH, obase = rbase * a.C + hh * DH;
#pragma unroll
    for (int n = 0; n < CH; ++n) {
      wa_dma16(a.q + ebase, voff[n], wa_lds_addr(Qs) + n * 1024);
      wa_dma16(a.k + ebase, voff[n], wa_lds_addr(Ks) + n * 1024);
      wa_dma16(a.v + ebase, voff[n], wa_lds_addr(Vs) + n * 1024);
      wa_dma16(a.dout + obase, voff_o[n], wa_lds_addr(Gs) + n * 1024);
    }
    const float my_lse = a.lse2[(bw * a.H + hh) * WA_N + lane];
    wa_wait_dma();
    lses[lane] = my_lse;
    // r re-derived per item from a value the compiler cannot see through: the 80 table offsets that depend on it would otherwise be
    // hoisted out of the item loop and held in registers
    int r = lane & 31;
    asm volatile("" : "+v"(r));
    // ================= query on the lane: delta, dQ
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = 32 * it + r;
      const float lse_i = lses[i];
      wa_f32x16 p[2], dp[2];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) p[jt][e] = 0.f, dp[jt][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          p[jt] = wa_mfma(wa_row_frag(Ks, ROWB, 32 * jt + r, ks, h), wa_row_frag(Qs, ROWB, i, ks, h), p[jt]);
          dp[jt] = wa_mfma(wa_row_frag(Vs, ROWB, 32 * jt + r, ks, h), wa_row_frag(Gs, ROWB, i, ks, h), dp[jt]);
        }
      }
      // delta_i = <dO_i, O_i> = sum_j P_ij dP_ij (O = P V): no need for O
      float del = 0.f;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 t = *reinterpret_cast<const float4*>(tab + i * WA_N + (((8 * jt + 2 * g + h) ^ (i & 15)) << 2));
          const float tt[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pe = __builtin_amdgcn_exp2f(fmaf(p[jt][4 * g + e], sc2, tt[e]) - lse_i);
            p[jt][4 * g + e] = pe;
            del = fmaf(pe, dp[jt][4 * g + e], del);
          }
        }
      del += __shfl_xor(del, 32);
      if (h == 0) dels[i] = del;
      wa_f32x16 dq;
#pragma unroll
      for (int e = 0; e < 16; ++e) dq[e] = 0.f;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          dq = wa_mfma(wa_tr_frag(Ks, ROWB, 32 * jt + 8 * g, h, tr_off),
                       wa_pack4(p[jt][4 * g] * (dp[jt][4 * g] - del), p[jt][4 * g + 1] * (dp[jt][4 * g + 1] - del),
                                p[jt][4 * g + 2] * (dp[jt][4 * g + 2] - del), p[jt][4 * g + 3] * (dp[jt][4 * g + 3] - del)), dq);
      wa_stage_out<DH>(Os, i, h, dq, a.scale);
      __builtin_amdgcn_sched_barrier(0);
    }
    wa_store_tile<DH>(Os, a.dq + ebase, voff, lane);
    // ================= key on the lane: dV, dK, bias gradient
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
      const int j = 32 * jt + r;
      wa_f32x16 dv, dk;
#pragma unroll
      for (int e = 0; e < 16; ++e) dv[e] = 0.f, dk[e] = 0.f;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        wa_f32x16 sa, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) sa[e] = 0.f, dp[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          sa = wa_mfma(wa_row_frag(Qs, ROWB, 32 * it + r, ks, h), wa_row_frag(Ks, ROWB, j, ks, h), sa);
          dp = wa_mfma(wa_row_frag(Gs, ROWB, 32 * it + r, ks, h), wa_row_frag(Vs, ROWB, j, ks, h), dp);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int i0 = 32 * it + 8 * g + 4 * h;
          const float4 l4 = *reinterpret_cast<const float4*>(lses + i0);
          const float4 d4 = *reinterpret_cast<const float4*>(dels + i0);
          const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
          float p[4], ds[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = i0 + e;
            const float t = tab[i * WA_N + ((((j >> 2) ^ (i & 15)) << 2) | (j & 3))];
            p[e] = __builtin_amdgcn_exp2f(fmaf(sa[4 * g + e], sc2, t) - ll[e]);
            ds[e] = p[e] * (dp[4 * g + e] - dd[e]);
            dT[it][jt][4 * g + e] += ds[e];
          }
          dv = wa_mfma(wa_tr_frag(Gs, ROWB, 32 * it + 8 * g, h, tr_off), wa_pack4(p[0], p[1], p[2], p[3]), dv);
          dk = wa_mfma(wa_tr_frag(Qs, ROWB, 32 * it + 8 * g, h, tr_off), wa_pack4(ds[0], ds[1], ds[2], ds[3]), dk);
        }
        __builtin_amdgcn_sched_barrier(0);   // (keeps the two query tiles' temporaries from being live together: 256 VGPRs, no spills)
      }
      // K rows / V rows 32 jt .. + 31 are dead from here on (the dQ pass is done, later key tiles read their own rows): the gradients
      // of exactly these rows are staged in their place
      wa_stage_out<DH>(Vs, j, h, dv, 1.f);
      wa_stage_out<DH>(Ks, j, h, dk, a.scale);
      __builtin_amdgcn_sched_barrier(0);
    }
    wa_store_tile<DH>(Vs, a.dv + ebase, voff, lane);
    wa_store_tile<DH>(Ks, a.dk + ebase, voff, lane);
  }
  // ---- the workgroup's bias-gradient partial: the four waves add up in the table's LDS, in wave order (fixed summation order)
  __syncthreads();
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float* p = tab + (32 * it + 8 * (e >> 2) + 4 * h + (e & 3)) * WA_N + 32 * jt + (lane & 31);
            *p = wv == 0 ? dT[it][jt][e] : *p + dT[it][jt][e];
          }
    }
    __syncthreads();
  }
  float* out = a.dtab_part + (long)blockIdx.x * (WA_N * WA_N);
#pragma unroll
  for (int n = 0; n < 4; ++n)
    *reinterpret_cast<float4*>(out + (threadIdx.x + 256 * n) * 4) = *reinterpret_cast<const float4*>(tab + (threadIdx.x + 256 * n) * 4);
}

// Slices of the batch per (window position, head): enough workgroups to fill the chip a few times over, a pair count that is a multiple
// of 8 (XCD map), at most one slice per sample -- and SHORT runs of items per workgroup.  The H workgroups that share a pair's 128-byte
// lines (48-byte head segments) start together and drift apart over a long run; round 5 measured the HBM-side bytes per launch against
// the run length (profiles/r05_window_attn_locality.json, 64 windows x 4 heads x 256 samples, samples per workgroup 32 .. 1):
//   forward  1.55 x algorithmic / 229 us at 32,  1.35 / 211 at 8,  1.12 / 225 at 4,  1.005 / 396 at 1 (three of four waves idle);
//   backward 1.69 / 499 at 32,  1.42 / 470 at 16,  1.25 / 558 at 8 (the per-workgroup bias-gradient partial and table load take over).
// Neither kernel is bound by those bytes: the fastest points are 8 samples per workgroup forward and 16 backward, chosen here.
static int wa_nsplit(int B, int nW, int H, bool bwd) {
  int ns = 1;
  long min_wgs = 2048;
  int run = bwd ? 16 : 8;     // samples per workgroup (4 waves: 4 resp. 2 items per wave)
  if (const char* e = MMK_DBG_ENV("MMK_WIN_MIN_WGS")) min_wgs = atol(e), run = B;   // experiment (debug-switch builds): the r5 sweep
  while (ns < B && ((long)nW * ns * H < min_wgs || ((nW * ns) & 7) != 0 || (B + ns - 1) / ns > run)) ns *= 2;
  return ns > B ? B : ns;
}

template <int DH>
static int wa_launch(const WinAttnArgs& a0, bool bwd, hipStream_t st) {
  WinAttnArgs a = a0;
  constexpr int TILE = WA_N * DH * 2;
  const size_t lds = bwd ? WA_TAB + 4 * (5 * TILE + 64 + 512) : WA_TAB + 4 * (3 * TILE + 64);
  const unsigned grid = (unsigned)(a.nW * a.nsplit * a.H);
  if (bwd) {
    MMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn_bwd_kernel<DH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfEvents pe(MMK_K_WIN_ATTN_BWD);   // dispatch-stamped start / stop (bench.py's per-kernel times), null when profiling is off
    hipExtLaunchKernelGGL((win_attn_bwd_kernel<DH>), dim3(grid), dim3(256), lds, st, pe.start, pe.stop, 0, a);
  } else {
    MMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(win_attn_fwd_kernel<DH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfEvents pe(MMK_K_WIN_ATTN_FWD);
    hipExtLaunchKernelGGL((win_attn_fwd_kernel<DH>), dim3(grid), dim3(256), lds, st, pe.start, pe.stop, 0, a);
  }
  MMK_LAUNCH_CHECK();
  return 0;
}

}  // namespace mmk

using namespace mmk;

extern "C" {

int mmk_win_attn_supported(int tokens, int dh, int c) { return tokens == WA_N && (dh == 24 || dh == 32) && c % dh == 0 && c % 8 == 0; }

// workgroups of the launch for (B samples, nW windows per sample, H heads) = rows of dtab_part the backward fills
int mmk_win_attn_blocks(int B, int nW, int H) { return nW * wa_nsplit(B, nW, H, true) * H; }

int mmk_win_attn_fwd(const void* q, const void* k, const void* v, const float* table, void* o, float* lse2, int B, int nW, int nWt, int H, int dh,
                     float scale, int img_h, int img_w, int shift, int ld, void* stream) {
  MMK_REQUIRE(q && k && v && table && o && lse2 && B > 0 && nW > 0 && H > 0, "win_attn_fwd: bad arguments");
  MMK_REQUIRE(dh == 24 || dh == 32, "win_attn: head dim must be 24 or 32");
  MMK_REQUIRE(nWt == 1 || nWt == nW, "win_attn: the table has one entry per head or one per (window position, head)");
  WinAttnArgs a = {};
  a.q = static_cast<const bf16_t*>(q); a.k = static_cast<const bf16_t*>(k); a.v = static_cast<const bf16_t*>(v);
  a.o = static_cast<bf16_t*>(o); a.lse2 = lse2; a.table = table;
  MMK_REQUIRE(img_w == 0 || (img_h > 0 && img_h % 8 == 0 && img_w % 8 == 0 && (img_h / 8) * (img_w / 8) == nW && shift >= 0 && shift < 8),
              "win_attn: a token map must be a whole number of 8 x 8 windows");
  a.B = B; a.nW = nW; a.nWt = nWt; a.H = H; a.C = H * dh; a.nsplit = wa_nsplit(B, nW, H, false); a.scale = scale;
  a.img_h = img_w > 0 ? img_h : 0; a.img_w = img_w; a.shift = img_w > 0 ? shift : 0;
  MMK_REQUIRE(ld == a.C || ld == 3 * a.C, "win_attn: q / k / v rows are C or 3 C elements apart");
  a.ld = ld;
  return dh == 24 ? wa_launch<24>(a, false, static_cast<hipStream_t>(stream)) : wa_launch<32>(a, false, static_cast<hipStream_t>(stream));
}

int mmk_win_attn_bwd(const void* q, const void* k, const void* v, const void* dout, const float* lse2, const float* table, void* dq, void* dk,
                     void* dv, float* dtab_part, int B, int nW, int nWt, int H, int dh, float scale, int img_h, int img_w, int shift, int ld,
                     void* stream) {
  MMK_REQUIRE(q && k && v && dout && lse2 && table && dq && dk && dv && dtab_part && B > 0 && nW > 0 && H > 0, "win_attn_bwd: bad arguments");
  MMK_REQUIRE(dh == 24 || dh == 32, "win_attn: head dim must be 24 or 32");
  MMK_REQUIRE(nWt == 1 || nWt == nW, "win_attn: the table has one entry per head or one per (window position, head)");
  WinAttnArgs a = {};
  a.q = static_cast<const bf16_t*>(q); a.k = static_cast<const bf16_t*>(k); a.v = static_cast<const bf16_t*>(v);
  a.dout = static_cast<const bf16_t*>(dout); a.lse2 = const_cast<float*>(lse2);
  a.table = table; a.dq = static_cast<bf16_t*>(dq); a.dk = static_cast<bf16_t*>(dk); a.dv = static_cast<bf16_t*>(dv); a.dtab_part = dtab_part;
  MMK_REQUIRE(img_w == 0 || (img_h > 0 && img_h % 8 == 0 && img_w % 8 == 0 && (img_h / 8) * (img_w / 8) == nW && shift >= 0 && shift < 8),
              "win_attn: a token map must be a whole number of 8 x 8 windows");
  a.B = B; a.nW = nW; a.nWt = nWt; a.H = H; a.C = H * dh; a.nsplit = wa_nsplit(B, nW, H, true); a.scale = scale;
  a.img_h = img_w > 0 ? img_h : 0; a.img_w = img_w; a.shift = img_w > 0 ? shift : 0;
  MMK_REQUIRE(ld == a.C || ld == 3 * a.C, "win_attn: q / k / v rows are C or 3 C elements apart");
  a.ld = ld;
  return dh == 24 ? wa_launch<24>(a, true, static_cast<hipStream_t>(stream)) : wa_launch<32>(a, true, static_cast<hipStream_t>(stream));
}

}  // extern "C"
