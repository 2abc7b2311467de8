// SURVEY 8(f1), second step: short-sequence self-attention for the encoders' transformer blocks (ViT-B/16: L = 197,
// own ViT / I-JEPA: L = 196 or 169, BERT: L = 77; head dim 64).  mmlearn's own block materialises softmax(QK^T) as a
// [B, h, L, L] tensor (mmlearn/modules/layers/attention.py:60-75); HF CLIP/BERT go through SDPA.  For L <= 256 one
// workgroup holds a whole (batch, head) problem on chip, so no online softmax is needed:
//
//   forward : K, V -> LDS by LDS-DMA as row-major images (128-B rows, 16-B chunks XOR-swizzled so that both the
//             ds_read_b128 row reads and the ds_read_b64_tr_b16 transposed reads are bank-conflict free); per 32-query
//             tile a wave computes S^T = K Q^T with the queries on the LANES (v_mfma_f32_32x32x16_bf16, Q fragments
//             straight from global memory), so max / sum over the keys are lane-local; the exponentiated accumulators
//             are then used directly as the B operand of O^T = V^T P^T (accumulator-as-operand, k order permuted
//             accordingly), with V^T fragments delivered by the hardware transpose read.
//             Persistent workgroups walk the (batch, head) items with the K / V images double-buffered.
//   backward: a streaming pass writes per-item records (lse * log2 e, delta = rowsum(dO . O)); the main kernel recomputes P
//             from Q, K and the LSE.  Two forms: a seven-product two-phase kernel with four resident images (no cross-wave
//             sums; any tile count), and -- whenever the tile count leaves a wave spare (L = 197, 77) -- a five-product
//             kernel in which the key waves hand their dS tile, transposed through LDS, to a dedicated dQ wave.
//             Dropout (BERT) is a counter-based mask regenerated from (seed, batch * head, query, key) in both passes.
//   HBM traffic = Q, K, V read once + O written once (forward), Q, K, V, dO, O in + dQ, dK, dV out (backward): both
//   are HBM-bound by bytes; HISTORY.md 5.2 has the measured distances to that bound.
#include <hip/hip_ext.h>
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace mmk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));

constexpr int ATT_DH = 64;  // head dimension (one 128-byte row of bf16)

// Workgroups per resident slot of the persistent kernels.  1 = every workgroup lives for the whole launch, which is the
// fastest when the GPU is ours alone; but a workgroup that starts late (CUs held by a concurrent kernel -- RCCL's
// all-reduce under DDP) then finishes a whole share late.  With a factor of 4 a share is a quarter of that, at the price
// of three more exposed prologues per slot (+2 % measured alone).  MMK_ATTN_GRID_FACTOR selects it; the default stays 1:
// DDP's bucketed all-reduce is active for a few per cent of the backward at 8 GPUs, less than the factor would cost.
static int att_grid_factor() {
  const char* e = getenv("MMK_ATTN_GRID_FACTOR");
  const int f = e ? atoi(e) : 1;
  return f < 1 ? 1 : (f > 64 ? 64 : f);
}

struct AttnArgs {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  bf16_t* out;   // [B, L, H, 64] contiguous
  float* lse;    // [B, H, L]
  long q_sb, q_sh, q_sl;  // element strides of [B, H, L, 64] views (last dim contiguous)
  long k_sb, k_sh, k_sl;
  long v_sb, v_sh, v_sl;
  int B, H, L;
  float scale;
  uint32_t seed_lo, seed_hi, drop_thr;  // attention dropout: drop (i, j) when its 16-bit draw < drop_thr (0 = off)
  float drop_scale;                     // 65536 / (65536 - drop_thr)
  const float* kbias;                   // MASK kernels: per-sample key bias records [B][ROWC] (mmk_attn_key_bias), base-2 logit units
  int causal;                           // MASK kernels: key j > query i is masked as well
};

// Key bias record of one sample (MASK kernels): ROWC floats, added to the base-2 logits of key j for EVERY head and query -- 0 for a
// key that is attended, KB_MASKED for a key behind a padding mask, -inf for j >= L.  KB_MASKED is finite on purpose (the HF convention of
// an additive finfo.min): a row whose keys are all masked averages V uniformly instead of producing NaN that a weight gradient summed
// over the batch would spread to every sample.
constexpr int ROWC = 256;
constexpr float KB_MASKED = -1e30f;

__device__ __forceinline__ float att_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// A copy of a per-lane value the optimiser cannot see through: address arithmetic built on it is redone where it is used
// instead of being hoisted out of the persistent item loop, where every hoisted value pins a VGPR for the whole kernel.
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// ---- LDS image of a [rows][64] bf16 tile: 128-byte rows, the 16-byte chunk index XORed with img_swz(row).
// img_swz = bit-reversed (row>>1)&7: rows 2t, 2t+1 keep their 128-B bank offset from the row parity, the 8 row pairs of
// a ds_read_b128 lane group land in 8 different chunks (row reads conflict-free), and rows i, i+2 of an aligned 4-row
// block sit in opposite 64-B halves (transposed reads conflict-free).
__device__ __forceinline__ int img_swz(int row) {
  const int t = row >> 1;
  return ((t & 1) << 2) | (t & 2) | ((t >> 2) & 1);
}

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to 1 KiB of LDS at the wave-uniform byte address
// `lds_addr`.  Written as inline asm (what the __builtin_amdgcn_global_load_lds builtin emits) so that the compiler does
// not track it: with the builtin, every later LDS read that might alias the destination gets a compiler-inserted
// s_waitcnt vmcnt(0), which would stall the compute of item n on the prefetch of item n+1.  The kernels order these
// writes themselves (s_waitcnt vmcnt(0) + barrier before the first read of an image); pieces are issued BEFORE any
// compiler-visible load of the same phase so the compiler's own vmcnt counts stay conservative.
__device__ __forceinline__ void lds_dma16(const void* sbase, uint32_t voff, uint32_t lds_addr) {
  // scalar 64-bit base + 32-bit per-lane byte offset: one VGPR of addressing per image instead of a pointer per piece
  const uint64_t pb = reinterpret_cast<uint64_t>(sbase);
  const uint64_t ps = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pb >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pb);
  sbase = reinterpret_cast<const void*>(ps);
  lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const char* p) {
  return (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)p);
}

// global [L][64] rows (element stride sl) -> image of LP rows; rows >= L repeat row L-1 (finite filler that the
// callers neutralise).  8 rows = 1 KiB per wave-instruction; the swizzle is applied to the SOURCE chunk.  `base`, `sl`,
// `L` must be wave-uniform.  Piece g of a wave advances the scalar base by 8 * nwaves rows; the per-lane offset (row
// 8 wave + lane/8, swizzled chunk) is the same for all of them because img_swz is periodic in 16 rows.
__device__ __forceinline__ void img_load(char* img, const bf16_t* base, long sl, int L, int LP, int wave, int nwaves,
                                         int lane) {
  const uint32_t img_addr = lds_addr_of(img);
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int row0 = 8 * wave_s + (lane >> 3);
  const int ch = (lane & 7) ^ img_swz(row0);
  const uint32_t voff = (uint32_t)(row0 * (int)sl + ch * 8) * 2u;
  for (int g = wave_s; g < LP / 8; g += nwaves) {
    const bf16_t* sbase = base + (long)(8 * (g - wave_s)) * sl;
    if (8 * g + 7 < L) {
      lds_dma16(sbase, voff, img_addr + g * 1024);
    } else {  // piece reaching past the last row: clamp per lane
      const int row = 8 * g + (lane >> 3);
      const uint32_t vo = (uint32_t)((min(row, L - 1) - 8 * (g - wave_s)) * (int)sl + ch * 8) * 2u;
      lds_dma16(sbase, vo, img_addr + g * 1024);
    }
  }
}

constexpr bool kAttnStagedDefault = true;   // (debug builds: MMK_ATTN_STAGED=0 / 1 selects the other form for A/B runs)

// ONE piece (rows 8 piece .. +7) of an image, for loaders whose piece indices are not congruent mod 16 rows (an odd number of
// loader waves): the per-lane offset is formed per piece.  Rows >= L repeat row L-1.
__device__ __forceinline__ void img_load_piece(char* img, const bf16_t* base, long sl, int L, int piece, int lane, int dst_piece = -1) {
  const int row = 8 * piece + (lane >> 3);
  const int ch = (lane & 7) ^ img_swz(row);   // (a destination that starts at a multiple of 16 rows keeps the source row's swizzle)
  const uint32_t voff = (uint32_t)(min(row, L - 1) * (int)sl + ch * 8) * 2u;
  lds_dma16(base, voff, lds_addr_of(img) + (dst_piece < 0 ? piece : dst_piece) * 1024);
}
// s_waitcnt vmcnt(n) for a run-time n in {0, 2, 4, 6} (wave-uniform)
__device__ __forceinline__ void wait_vmem_upto(int n) {
  if (n >= 6) __builtin_amdgcn_s_waitcnt(0x0F76);
  else if (n >= 4) __builtin_amdgcn_s_waitcnt(0x0F74);
  else if (n >= 2) __builtin_amdgcn_s_waitcnt(0x0F72);
  else __builtin_amdgcn_s_waitcnt(0x0F70);
}

// MFMA operands out of an image.  k along the 64 COLUMNS: lane (r, h) takes row `row`, elements 16kk + 8h .. +7, one
// ds_read_b128 at  row * 128 + (((2 kk + h) ^ img_swz(row)) << 4).  k along the ROWS (transposed use of the same image):
// for the 32x32x16 A operand X^T[c][k] with c = 32 ct + (lane & 31) and the accumulator-as-operand k order, the lane
// needs column c of rows r0 .. r0+3 and r0+8 .. r0+11 (r0 = 16 s + 4 h + tile base): two ds_read_b64_tr_b16, for which
// lane 4q + p of each 16-lane group supplies the address of row r0 + q, columns 4p .. 4p+3 of the group's 4 x 16 block
// (chunk 4 ct + 2 (lane>>4 & 1) + (p >> 1), byte 8 (p & 1)) and receives column (lane & 15) of the four rows.
// Per-lane byte offsets into an image that do not depend on the 32-row tile: a tile adds 4096 B, the second k-step of
// a transposed fragment (rows +16) adds 2048 B, because img_swz only looks at (row >> 1) & 7.  Computing them once
// keeps the integer address arithmetic out of the MFMA loops (the loops were VALU-issue bound without this).
struct ImgLane {
  int row[4];    // row fragment kk of row 32t + (lane & 31)             = img + 4096 t + row[kk]
  int tr[2][2];  // transposed half u of k-step s, column tile ct        = img + 4096 t + 2048 s + tr[u][ct]
};
__device__ __forceinline__ ImgLane img_lane(int lane) {
  ImgLane o;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) o.row[kk] = r * 128 + (((2 * kk + h) ^ img_swz(r)) << 4);
  const int li = lane & 15, q = li >> 2, p = li & 3;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int row = 4 * h + 8 * u + q;
      const int ch = 4 * ct + 2 * ((lane >> 4) & 1) + (p >> 1);
      o.tr[u][ct] = row * 128 + ((ch ^ img_swz(row)) << 4) + 8 * (p & 1);
    }
  return o;
}
__device__ __forceinline__ bf16x8 lds_row_frag(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 lds_tr_frag(const char* img, const ImgLane& il, int tile_s_off, int ct) {
  typedef short s4 __attribute__((ext_vector_type(4)));
  const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(img + tile_s_off + il.tr[0][ct]));
  const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(img + tile_s_off + il.tr[1][ct]));
  typedef short s8 __attribute__((ext_vector_type(8)));
  s8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return __builtin_bit_cast(bf16x8, f);
}

constexpr int STAGE_BYTES = 32 * 128;  // one wave's [32 rows][64] bf16 staging tile, 16-B chunk index ^= row & 7

// acc[dt][e] = X^T[d][row], d = 32dt + 8(e>>2) + 4h + (e&3), the row on the lane  ->  rows of a [.., 64] bf16 tensor.
// Staged through a per-wave LDS tile so that global stores are whole 128-B rows (8 lanes x 16 B): a row-per-lane store
// touches 32 cache lines per instruction and is issue-bound.  ROWS = 32 stages the tile at once (4 KiB per wave),
// ROWS = 16 in two halves (2 KiB per wave, for the configurations whose images leave no more LDS).
template <int ROWS = 32>
__device__ __forceinline__ void store_rows_staged(char* stage, bf16_t* dst, long row_stride, int row0, int nrows_valid,
                                                  const f32x16 (&acc)[2], float mul, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int half = 0; half < 32 / ROWS; ++half) {
    const int rs = r - half * ROWS;  // row inside the staged slab
    if (ROWS == 32 || (rs >= 0 && rs < ROWS)) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          bf16x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = (bf16_t)(acc[dt][4 * q4 + e] * mul);
          *reinterpret_cast<bf16x4*>(stage + rs * 128 + (((dt * 4 + q4) ^ (rs & 7)) << 4) + 8 * h) = w;
        }
    }
#pragma unroll
    for (int it = 0; it < ROWS / 8; ++it) {
      const int row = it * 8 + (lane >> 3), ch = lane & 7;
      const bf16x8 val = *reinterpret_cast<const bf16x8*>(stage + row * 128 + ((ch ^ (row & 7)) << 4));
      const int grow = row0 + half * ROWS + row;
      if (grow < nrows_valid) *reinterpret_cast<bf16x8*>(dst + (long)grow * row_stride + ch * 8) = val;
    }
  }
}

// The same store for the five-product backward (32-row slabs), which also feeds column sums: cs8 accumulates, per lane, the sums of
// its 8 columns (16-byte chunk lane & 7) over the rows it stored -- of the ROUNDED values, i.e. exactly what a later dY.sum(0) over
// the stored tensor would add up (the bias gradient of a fused QKV projection).  cs_flush reduces them over the wave and writes
// the 64 sums.
__device__ __forceinline__ void store_rows_staged_cs(char* stage, bf16_t* dst, long row_stride, int row0, int nrows_valid,
                                                     const f32x16 (&acc)[2], float mul, int lane, float (&cs8)[8]) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      bf16x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (bf16_t)(acc[dt][4 * q4 + e] * mul);
      *reinterpret_cast<bf16x4*>(stage + r * 128 + (((dt * 4 + q4) ^ (r & 7)) << 4) + 8 * h) = w;
    }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = it * 8 + (lane >> 3), ch = lane & 7;
    const bf16x8 val = *reinterpret_cast<const bf16x8*>(stage + row * 128 + ((ch ^ (row & 7)) << 4));
    const int grow = row0 + row;
    if (grow < nrows_valid) {
      *reinterpret_cast<bf16x8*>(dst + (long)grow * row_stride + ch * 8) = val;
#pragma unroll
      for (int e = 0; e < 8; ++e) cs8[e] += (float)val[e];
    }
  }
}

// Lanes ch + 8 k (k = lane >> 3) hold partial sums of the same 8 columns 8 ch .. 8 ch + 7.  One trip through the wave's staging tile
// (free again once its rows are read back; a wave's LDS operations execute in order) instead of three dependent levels of cross-lane
// exchanges: lane (ch, k) leaves its 8 sums at [k][8 ch ..], lane c adds up column c over the 8 k and stores it -- one coalesced
// 256-byte store.  cs (wave-uniform) may be null: nothing is wanted.
__device__ __forceinline__ void cs_flush(char* stage, const float (&cs8)[8], int lane, float* cs) {
  if (!cs) return;
  float* st = reinterpret_cast<float*>(stage);
  *reinterpret_cast<float4*>(st + (lane >> 3) * 64 + (lane & 7) * 8) = make_float4(cs8[0], cs8[1], cs8[2], cs8[3]);
  *reinterpret_cast<float4*>(st + (lane >> 3) * 64 + (lane & 7) * 8 + 4) = make_float4(cs8[4], cs8[5], cs8[6], cs8[7]);
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = st[k * 64 + lane];
  cs[lane] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

// v_max3_f32 as one instruction: fmaxf on MFMA results makes the compiler canonicalise each operand first (v_max x, x)
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// rows of a wave's output staging slab in the forward: 32, or 16 where the key bias records would not fit beside 32 (L > 224 fills
// the 160 KiB with its four images and eight 4-KiB slabs)
constexpr int fwd_stage_rows(int NT, bool MASK) { return (MASK && NT == 8) ? 16 : 32; }

// One 32-query tile of one (batch, head): queries i = 32t + (lane&31) against all keys of the K / V images.
// S^T tile jt: acc[reg] = sum_d K[j][d] Q[i][d],  j = 32jt + (reg&3) + 8(reg>>2) + 4h, the query on the LANE, so the
// row maximum and sum are lane-local.  ONEPASS keeps all NT score tiles in registers (16 NT VGPRs, NT <= 7 at two waves
// per SIMD); otherwise QK^T is formed twice (pass 1: maximum, pass 2: exponentials + PV).
template <int NT, bool DROP, bool ONEPASS, bool MASK>
__device__ __forceinline__ void attn_fwd_tile(const AttnArgs& a, const char* Ks, const char* Vs, char* stage, const float* kb,
                                              const ImgLane& il, const bf16x8 (&qf)[4], int bh, int t, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const int b = bh / a.H, hh = bh % a.H;
  const int i = t * 32 + r;  // this lane's query row
  const float sl2 = a.scale * 1.4426950408889634f;  // scale > 0: the row maximum commutes with the scaling
  const uint32_t dkey = DROP ? drop_key(a.seed_lo, a.seed_hi, (uint32_t)bh) : 0u;
  // raw q.k with keys beyond L (last tile only) = -inf; MASK: base-2 logits  sl2 q.k + kb[j]  (the record holds -inf beyond L)
  auto score_tile = [&](int jt) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(Ks + jt * 4096 + il.row[kk]), qf[kk], acc, 0, 0, 0);
    if (MASK) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 b4 = *reinterpret_cast<const float4*>(kb + jt * 32 + 8 * g4 + 4 * h);
        acc[4 * g4 + 0] = fmaf(acc[4 * g4 + 0], sl2, b4.x);
        acc[4 * g4 + 1] = fmaf(acc[4 * g4 + 1], sl2, b4.y);
        acc[4 * g4 + 2] = fmaf(acc[4 * g4 + 2], sl2, b4.z);
        acc[4 * g4 + 3] = fmaf(acc[4 * g4 + 3], sl2, b4.w);
      }
      if (a.causal) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h > i) acc[e] = -INFINITY;
      }
    } else if (jt == NT - 1 && a.L < 32 * NT) {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= a.L) acc[e] = -INFINITY;
    }
    return acc;
  };
  const float mul = MASK ? 1.f : sl2;   // what is left to apply to a score tile
  float sum = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  // exponentials of one score tile, then O^T[d][i] += sum_j V^T[d][j] P^T[j][i]: A = V^T fragment (transposed read),
  // B = the accumulator tile itself: the B fragment of k-step s (s = 0,1) is regs 8s..8s+7, whose element jj is row
  // 16s + 8(jj>>2) + 4h + (jj&3) of the tile
  auto pv_tile = [&](int jt, f32x16 x, float nm2) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      x[e] = att_exp2(fmaf(x[e], mul, nm2));
      sum += x[e];
    }
    if (DROP) {  // the row sum above is the softmax denominator; dropped weights only leave the PV product
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const uint32_t w = drop_word(dkey, i, (jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) >> 1);
        if ((w & 0xFFFFu) < a.drop_thr) x[e] = 0.f;
        if ((w >> 16) < a.drop_thr) x[e + 1] = 0.f;
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 pf;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) pf[jj] = (bf16_t)x[8 * s + jj];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(Vs, il, jt * 4096 + s * 2048, dt), pf, o[dt], 0, 0, 0);
    }
  };
  float m = -INFINITY;
  if (ONEPASS) {
    f32x16 sc[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) sc[jt] = score_tile(jt);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int e = 0; e < 16; e += 2) m = max3(m, sc[jt][e], sc[jt][e + 1]);
    m = fmaxf(m, __shfl_xor(m, 32));
    const float nm2 = -m * mul;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) pv_tile(jt, sc[jt], nm2);
  } else {
#pragma unroll 1
    for (int jt = 0; jt < NT; ++jt) {
      const f32x16 x = score_tile(jt);
#pragma unroll
      for (int e = 0; e < 16; e += 2) m = max3(m, x[e], x[e + 1]);
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    const float nm2 = -m * mul;
#pragma unroll 1
    for (int jt = 0; jt < NT; ++jt) pv_tile(jt, score_tile(jt), nm2);
  }
  sum += __shfl_xor(sum, 32);
  // ---- epilogue: o[dt][reg] = O^T[d][i], d = 32dt + (reg&3) + 8(reg>>2) + 4h ; normalise per lane, store whole rows
  const float inv = (DROP ? a.drop_scale : 1.f) / sum;
  store_rows_staged<fwd_stage_rows(NT, MASK)>(stage, a.out + ((long)b * a.L * a.H + hh) * ATT_DH, (long)a.H * ATT_DH, t * 32, a.L, o, inv, lane);
  if (i < a.L && h == 0) a.lse[((long)b * a.H + hh) * a.L + i] = (m * mul + log2f(sum)) * 0.6931471805599453f;
}

// Q fragments straight from global: lane (r, h) of the wave owning tile t needs Q[i][16kk + 8h .. +8]
__device__ __forceinline__ void attn_load_q(const AttnArgs& a, int bh, int t, int lane, bf16x8 (&qf)[4]) {
  const int b = bh / a.H, hh = bh % a.H;
  const int ic = min(t * 32 + (lane & 31), a.L - 1);
  const bf16_t* qrow = a.q + b * a.q_sb + hh * a.q_sh + (long)ic * a.q_sl + 8 * (lane >> 5);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk] = *reinterpret_cast<const bf16x8*>(qrow + 16 * kk);
}

// s_waitcnt vmcnt(0) as the BUILTIN (expcnt / lgkmcnt fields left at their maxima): unlike an asm statement the
// compiler's wait-count bookkeeping sees it, so it knows every earlier load has landed and inserts no vmcnt of its own
// in the compute phase -- vmcnt is in-order, and any such wait would also wait for the prefetch issued after it.
__device__ __forceinline__ void wait_vmem_all() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// Persistent workgroups: NT <= NW, wave w owns query tile w of every (batch, head) item the workgroup walks.  The K / V
// images are double-buffered: the LDS-DMA of item n+1 and its Q fragments are in flight while item n is computed, so
// the HBM stream never stops behind a compute phase (one barrier per item).
template <int NT, int NW, bool DROP, bool MASK>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const AttnArgs a) {
  constexpr int LP = 32 * NT;
  constexpr int IMG = LP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nitems = a.B * a.H;
  constexpr int STG = fwd_stage_rows(NT, MASK) * 128;                         // one wave's staging slab
  float* kbs = reinterpret_cast<float*>(smem + 4 * IMG + NW * STG);           // MASK: [2 buffers][ROWC] key bias records
  auto issue_kv = [&](int bh, int buf) {
    const int b = bh / a.H, hh = bh % a.H;
    img_load(smem + (2 * buf) * IMG, a.k + b * a.k_sb + hh * a.k_sh, a.k_sl, a.L, LP, wave, NW, lane);
    img_load(smem + (2 * buf + 1) * IMG, a.v + b * a.v_sb + hh * a.v_sh, a.v_sl, a.L, LP, wave, NW, lane);
    if (MASK && __builtin_amdgcn_readfirstlane(wave) == NW - 1)   // the sample's record rides with the images (one 1-KiB piece)
      lds_dma16(a.kbias + (long)b * ROWC, (uint32_t)lane * 16u, lds_addr_of(reinterpret_cast<const char*>(kbs + buf * ROWC)));
  };
  char* stage = smem + 4 * IMG + wave * STG;
  const ImgLane il = img_lane(lane);
  int item = blockIdx.x;
  bf16x8 qn[4];  // Q fragments of the item whose images are in flight
  if (item < nitems) {
    if (wave < NT) attn_load_q(a, item, wave, lane, qn);
    asm volatile("" ::: "memory");
    issue_kv(item, 0);
  }
  for (int n = 0; item < nitems; item += gridDim.x, ++n) {
    wait_vmem_all();  // this wave's DMA pieces and Q fragments of `item` have landed
    __syncthreads();  // images of `item` complete; every wave is done with the buffers of the previous item
    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = qn[kk];
    const int next = item + gridDim.x;
    if (next < nitems) {
      // prefetch of the next item: compiler-visible Q loads FIRST, then the LDS-DMA pieces it does not track
      if (wave < NT) attn_load_q(a, next, wave, lane, qn);
      asm volatile("" ::: "memory");  // keeps the Q loads from being sunk to their first use
      issue_kv(next, (n + 1) & 1);
    }
    if (wave < NT)
      attn_fwd_tile<NT, DROP, (DROP ? NT <= 6 : NT <= 7), MASK>(a, smem + (2 * (n & 1)) * IMG, smem + (2 * (n & 1) + 1) * IMG, stage,
                                                                kbs + (n & 1) * ROWC, il, qf, item, wave, lane);
  }
}

template <int NT, bool DROP, bool MASK>
static int launch_attn_fwd(const AttnArgs& a, hipStream_t st) {
  constexpr int LP = 32 * NT;
  constexpr int NW = NT <= 4 ? 4 : 8;
  constexpr int bytes = 4 * LP * 128 + NW * fwd_stage_rows(NT, MASK) * 128 + (MASK ? 2 * ROWC * 4 : 0);
  static_assert(bytes <= 160 * 1024, "LDS budget");
  auto kern = attn_fwd_kernel<NT, NW, DROP, MASK>;
  KernelSetup ks;   // LDS opt-in, occupancy and CU count of this kernel on the current device
  if (int rc = kernel_setup(reinterpret_cast<const void*>(kern), 64 * NW, bytes, &ks)) return rc;
  const int cus = ks.cus, wgs_per_cu = ks.wgs_per_cu;
  const int grid = std::min(a.B * a.H, cus * wgs_per_cu * att_grid_factor());
  ProfEvents pe(MMK_K_ATTN_FWD);
  hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), bytes, st, pe.start, pe.stop, 0, a);
  MMK_LAUNCH_CHECK();
  return 0;
}


// ------------------------------------------------------------------------------------------------ backward
//   P = exp2(c S - lse2) is recomputed from Q, K and the forward's LSE; delta_i = sum_d dO_id O_id.
//   Q, K, V, dO of one (batch, head) live in LDS as four images (read once from HBM); two phases, no cross-wave sums:
//     phase 1 (a wave owns 32 KEYS):    S = Q K^T, dP = dO V^T with the key on the lane; the accumulators, turned into
//                                       P and dS = P (dP - delta), are the B operands of dV^T += dO^T P and
//                                       dK^T += Q^T dS (A operands = transposed reads of the dO / Q images);
//     phase 2 (a wave owns 32 QUERIES): S^T = K Q^T, dP^T = V dO^T with the query on the lane, dS^T is the B operand of
//                                       dQ^T += K^T dS^T (transposed reads of the K image).
//   Seven MFMA products per tile pair instead of five (S and dP are formed in both orientations) buys a kernel with no
//   atomics, no dS round trip through LDS and one barrier; the pass is HBM-bound (5 reads + 3 writes of [L, 64]).
struct AttnBwdArgs {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  const bf16_t* o;     // [B, L, H, 64] contiguous
  const bf16_t* dout;  // [B, L, H, 64] contiguous
  const float* lse;    // [B, H, L] (natural log)
  float* delta;        // [B * H][2][256] workspace: lse2 and delta rows, filled by attn_delta_kernel
  bf16_t* dq;          // [B, L, H, 64] views: element strides g_sb (batch), g_sl (row); heads 64 apart.  A packed
  bf16_t* dk;          // [B, L, 3, H, 64] gradient buffer for a fused QKV projection is g_sl = 3 H 64 with the three
  bf16_t* dv;          // pointers H 64 elements apart
  long g_sb, g_sl;
  long q_sb, q_sh, q_sl;
  long k_sb, k_sh, k_sl;
  long v_sb, v_sh, v_sl;
  int B, H, L;
  float scale;
  uint32_t seed_lo, seed_hi, drop_thr;
  float drop_scale;
  float* cs;  // optional [B * ceil(L / 32)][3][H][64] f32: per 32-row tile column sums of the stored dq / dk / dv (packed layout)
  const float* kbias;  // MASK kernels: the forward's key bias records [B][ROWC]
  int causal;          // MASK kernels
  unsigned long long* stamps;  // debugging (MMK_ATTN_STAMPS): shader-clock stamps of workgroup 0, 16 per item, first 32 items
};

// where the column sums of tile `tile` of (batch b, head hh), part 0 = dq / 1 = dk / 2 = dv, go (nullptr: not wanted)
__device__ __forceinline__ float* cs_slot(const AttnBwdArgs& a, int b, int hh, int tile, int part) {
  return a.cs ? a.cs + (((long)b * ((a.L + 31) >> 5) + tile) * 3 + part) * ((long)a.H * ATT_DH) + hh * ATT_DH : nullptr;
}

// Row constants of the backward, one 2-KiB record per (batch, head):  ws[bh][0][l] = lse[bh][l] * log2(e) (+inf for
// l >= L, which makes P = 0 on padded rows), ws[bh][1][l] = delta = sum_d dO[b, l, h, d] * O[b, l, h, d] (0 for l >= L).
// 256 floats per row so that the main kernel fetches a record with two 1-KiB LDS-DMA pieces whatever L is.
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ g,
                                                         const float* __restrict__ lse, float* __restrict__ ws, int B,
                                                         int H, int L) {
  const long nrows = (long)B * L * H;
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nrows * 8; c += (long)gridDim.x * 256) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(o + c * 8);  // 8 lanes per 128-byte row, fully coalesced
    const bf16x8 y = *reinterpret_cast<const bf16x8*>(g + c * 8);
    float part = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) part += (float)x[e] * (float)y[e];
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    part += __shfl_xor(part, 4);
    if ((c & 7) == 0) {
      const long row = c >> 3;  // (b * L + l) * H + h
      const int hh = (int)(row % H);
      const long bl = row / H;
      const int l = (int)(bl % L);
      const long b = bl / L;
      ws[((b * H + hh) * 2 + 1) * ROWC + l] = part;
    }
  }
  const long nbh = (long)B * H;
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nbh * ROWC; c += (long)gridDim.x * 256) {
    const long bh = c / ROWC;
    const int l = (int)(c % ROWC);
    ws[(bh * 2) * ROWC + l] = l < L ? lse[bh * L + l] * 1.4426950408889634f : INFINITY;
    if (l >= L) ws[(bh * 2 + 1) * ROWC + l] = 0.f;
  }
}

// P of one (query i, key j) pair in the backward: exp2 of the base-2 logit minus the row's lse2.  MASK: the key bias is added to the
// scaled score FIRST and the lse2 subtracted from the rounded sum -- a row whose keys are all masked has logits and lse2 of about
// KB_MASKED, where  sl2 s + (kb - lse2)  would lose the bias altogether and return exp2(sl2 s), possibly huge; this order gives P = 1 on
// such a row (finite; the row's gradients mean nothing, but they do not poison the batch-summed weight gradients).
template <bool MASK>
__device__ __forceinline__ float bwd_prob(float s, float sl2, float kbj, float lse2, bool jvalid, int causal, int j, int i) {
  if (!MASK) return jvalid ? att_exp2(fmaf(s, sl2, -lse2)) : 0.f;
  const float p = att_exp2(fmaf(s, sl2, kbj) - lse2);
  return (jvalid && !(causal && j > i)) ? p : 0.f;
}

// Persistent workgroups, NT <= NW: wave w owns key tile w in phase 1 and query tile w in phase 2 of every (batch, head)
// item its workgroup walks.  Four image buffers (Q, dO, K, V) and no idle HBM phase:
//   barrier A  Q, dO images of item n and its delta / lse2 rows are in LDS; K, V buffers are free
//              -> issue the LDS-DMA of K, V (n);  PHASE 1 (n) on the Q / dO images with K_j, V_j fragments that were
//                 prefetched into registers;  pick up this wave's Q_i, dO_i fragments for phase 2
//   barrier B  K, V images of item n landed; Q, dO buffers are free
//              -> prefetch item n+1: K_j, V_j fragments into registers, LDS-DMA of Q, dO (n+1) and of its lse2 /
//                 delta record into the other row-constant buffer;  PHASE 2 (n) on the K / V images
// The lse2 / delta records (delta_i = sum_d dO_id O_id) come from attn_delta_kernel, one streaming pass before this one.
// Compiler-visible global loads are always issued BEFORE the untracked LDS-DMA pieces of the same window and are
// consumed only after the next barrier's s_waitcnt vmcnt(0), so no compiler-inserted vmcnt lands inside a phase.
template <int NT, int NW, bool DROP, bool MASK>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_bwd_kernel(const AttnBwdArgs a) {
  constexpr int LP = 32 * NT;
  constexpr int IMG = LP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* Gs = Qs + IMG;  // dO
  char* Ks = Gs + IMG;
  char* Vs = Ks + IMG;
  constexpr int SROWS = NT <= 7 ? 32 : 16;  // staging slab rows per wave (LDS budget at NT = 8)
  constexpr int REC = MASK ? 3 : 2;                  // row-constant records per buffer
  float* rowc = reinterpret_cast<float*>(Vs + IMG);  // [2 buffers][lse2 | delta | (MASK) key bias][ROWC]
  char* stage = reinterpret_cast<char*>(rowc + 2 * REC * ROWC) + (threadIdx.x >> 6) * (SROWS * 128);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const ImgLane il = img_lane(lane);
  const long osl = (long)a.H * ATT_DH;
  const float sl2 = a.scale * 1.4426950408889634f;
  const int nitems = a.B * a.H;
  const bool active = wave < NT;  // wave-uniform: this wave owns tile `wave`

  bf16x8 kf[4], vf[4];        // K_j, V_j fragments of the item entering phase 1

  auto obase_of = [&](int bh) { return ((long)(bh / a.H) * a.L * a.H + (bh % a.H)) * ATT_DH; };
  // compiler-visible prefetch loads of item bh (K_j, V_j fragments)
  // (scalar base + 32-bit per-lane byte offset: no 64-bit per-lane pointers to keep alive across the item loop)
  auto load_regs = [&](int bh) {
    const int b = bh / a.H, hh = bh % a.H;
    if (active) {
      const int lo = opaque(lane);
      const int jc = min(wave * 32 + (lo & 31), a.L - 1);
      const uint32_t koff = (uint32_t)(jc * (int)a.k_sl + 8 * (lo >> 5)) * 2u;
      const uint32_t voff = (uint32_t)(jc * (int)a.v_sl + 8 * (lo >> 5)) * 2u;
      const char* kbase = reinterpret_cast<const char*>(a.k + b * a.k_sb + hh * a.k_sh);
      const char* vbase = reinterpret_cast<const char*>(a.v + b * a.v_sb + hh * a.v_sh);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        kf[kk] = *reinterpret_cast<const bf16x8*>(kbase + koff + 32 * kk);
        vf[kk] = *reinterpret_cast<const bf16x8*>(vbase + voff + 32 * kk);
      }
    }
    asm volatile("" ::: "memory");  // keep the loads here (not sunk to their first use)
  };
  auto issue_qg = [&](int bh, int buf) {  // + the item's row-constant record (two pieces, waves 0 and 1)
    const int b = bh / a.H, hh = bh % a.H;
    const int lo = opaque(lane);
    img_load(Qs, a.q + b * a.q_sb + hh * a.q_sh, a.q_sl, a.L, LP, wave, NW, lo);
    img_load(Gs, a.dout + obase_of(bh), osl, a.L, LP, wave, NW, lo);
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    if (wave_s < 2)
      lds_dma16(a.delta + ((long)bh * 2 + wave_s) * ROWC, (uint32_t)lo * 16u,
                lds_addr_of(reinterpret_cast<const char*>(rowc + (buf * REC + wave_s) * ROWC)));
    if (MASK && wave_s == 2)
      lds_dma16(a.kbias + (long)b * ROWC, (uint32_t)lo * 16u, lds_addr_of(reinterpret_cast<const char*>(rowc + (buf * REC + 2) * ROWC)));
  };
  auto issue_kv = [&](int bh) {
    const int b = bh / a.H, hh = bh % a.H;
    const int lo = opaque(lane);
    img_load(Ks, a.k + b * a.k_sb + hh * a.k_sh, a.k_sl, a.L, LP, wave, NW, lo);
    img_load(Vs, a.v + b * a.v_sb + hh * a.v_sh, a.v_sl, a.L, LP, wave, NW, lo);
  };

  int item = blockIdx.x;
  if (item < nitems) {
    load_regs(item);
    issue_qg(item, 0);
    wait_vmem_all();
  }
  // The s_waitcnt vmcnt(0) for a window's DMA sits at the END of the following MFMA loop, before that phase's output
  // stores: the pieces had the whole phase to land, and the stores stay in flight across the barrier.
  for (int n = 0; item < nitems; item += gridDim.x, ++n) {
    const float* lse2s = rowc + (n & 1) * REC * ROWC;
    const float* dls = lse2s + ROWC;
    const float* kbs = dls + ROWC;   // MASK only
    const long gbase = (long)(item / a.H) * a.g_sb + (long)(item % a.H) * ATT_DH;
    const uint32_t dkey = DROP ? drop_key(a.seed_lo, a.seed_hi, (uint32_t)item) : 0u;
    __syncthreads();  // ---- barrier A
    issue_kv(item);
    bf16x8 qf[4], gf[4];
    f32x16 acc1[2], acc2[2];  // phase 1: dK^T, dV^T;  phase 2: dQ^T (acc1)
    if (active) {
      // ---------------- phase 1: dK, dV of key tile `wave` (the key on the lane)
      const int j = wave * 32 + r;
      const bool jvalid = j < a.L;
      const float kbj = MASK ? kbs[j] : 0.f;   // this lane's key
      f32x16 (&dkt)[2] = acc1;
      f32x16 (&dvt)[2] = acc2;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) dkt[dt][e] = dvt[dt][e] = 0.f;
#pragma unroll 1
      for (int it = 0; it < NT; ++it) {
        f32x16 sc, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[e] = dp[e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(Qs + it * 4096 + il.row[kk]), kf[kk], sc, 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(Gs + it * 4096 + il.row[kk]), vf[kk], dp, 0, 0, 0);
        bf16x8 pf[2], df[2];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float4 l2 = *reinterpret_cast<const float4*>(lse2s + it * 32 + 8 * g4 + 4 * h);
          const float4 dl = *reinterpret_cast<const float4*>(dls + it * 32 + 8 * g4 + 4 * h);
          const float l2v[4] = {l2.x, l2.y, l2.z, l2.w};
          const float dlv[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            const int e = 4 * g4 + e4;
            const float p = bwd_prob<MASK>(sc[e], sl2, kbj, l2v[e4], jvalid, a.causal, j, it * 32 + 8 * g4 + 4 * h + e4);
            float keep = 1.f;
            if (DROP) {
              const uint32_t w = drop_word(dkey, it * 32 + 8 * g4 + 4 * h + e4, j >> 1);
              keep = ((w >> (16 * (j & 1))) & 0xFFFFu) < a.drop_thr ? 0.f : a.drop_scale;
            }
            pf[e >> 3][e & 7] = (bf16_t)(DROP ? p * keep : p);
            df[e >> 3][e & 7] = (bf16_t)(p * ((DROP ? dp[e] * keep : dp[e]) - dlv[e4]));
          }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(Gs, il, it * 4096 + s * 2048, dt), pf[s], dvt[dt], 0, 0, 0);
            dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(Qs, il, it * 4096 + s * 2048, dt), df[s], dkt[dt], 0, 0, 0);
          }
      }
    }
    wait_vmem_all();  // K, V pieces of this item (issued before phase 1)
    if (active) {
      store_rows_staged<SROWS>(stage, a.dk + gbase, a.g_sl, wave * 32, a.L, acc1, a.scale, opaque(lane));
      store_rows_staged<SROWS>(stage, a.dv + gbase, a.g_sl, wave * 32, a.L, acc2, 1.f, opaque(lane));
      // this wave's query-side fragments for phase 2, before the Q / dO buffers are handed to the next item
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        qf[kk] = lds_row_frag(Qs + wave * 4096 + il.row[kk]);
        gf[kk] = lds_row_frag(Gs + wave * 4096 + il.row[kk]);
      }
    }
    __syncthreads();  // ---- barrier B
    const int next = item + gridDim.x;
    if (next < nitems) {
      load_regs(next);
      issue_qg(next, (n + 1) & 1);
    }
    if (active) {
      // ---------------- phase 2: dQ of query tile `wave` (the query on the lane)
      const int i = wave * 32 + r;
      const float l2 = lse2s[i], dl = dls[i];
      f32x16 (&dqt)[2] = acc1;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) dqt[dt][e] = 0.f;
#pragma unroll 1
      for (int jt = 0; jt < NT; ++jt) {
        f32x16 sc, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[e] = dp[e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(Ks + jt * 4096 + il.row[kk]), qf[kk], sc, 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(Vs + jt * 4096 + il.row[kk]), gf[kk], dp, 0, 0, 0);
        if (DROP) {
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            const uint32_t w = drop_word(dkey, i, (jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) >> 1);
            dp[e] *= (w & 0xFFFFu) < a.drop_thr ? 0.f : a.drop_scale;
            dp[e + 1] *= (w >> 16) < a.drop_thr ? 0.f : a.drop_scale;
          }
        }
        bf16x8 df[2];
        if (MASK) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int jk = jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            df[e >> 3][e & 7] = (bf16_t)(bwd_prob<true>(sc[e], sl2, kbs[jk], l2, true, a.causal, jk, i) * (dp[e] - dl));
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) df[e >> 3][e & 7] = (bf16_t)(att_exp2(fmaf(sc[e], sl2, -l2)) * (dp[e] - dl));
        }
        if (jt == NT - 1 && a.L < LP) {  // keys beyond L (finite filler rows): dS = 0
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= a.L) df[e >> 3][e & 7] = (bf16_t)0.f;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
            dqt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(Ks, il, jt * 4096 + s * 2048, dt), df[s], dqt[dt], 0, 0, 0);
      }
    }
    wait_vmem_all();  // Q, dO pieces and the register prefetch of the next item
    if (active) store_rows_staged<SROWS>(stage, a.dq + gbase, a.g_sl, wave * 32, a.L, acc1, a.scale, opaque(lane));
  }
}

#ifdef MMK_DEBUG_SWITCHES
// EXPERIMENT (debug-switch builds only, MMK_ATTN_SPLIT=1; measured, not kept: 18-35 % SLOWER than the five-product kernel at
// L = 197 / 77 / 169, profiles/r04_attn_bwd.json -- two more exponential passes and twice the operand reads cost more than the step
// barriers of the one-kernel form).
// ---- the two-kernel form: the seven-product kernel's two phases as separate launches, so that no wave ever
// waits at a barrier for a wave in the other role.  PH = 1: dK, dV of an item (a wave owns 32 keys, K_j / V_j fragments in
// registers, sweeps the query tiles of the Q / dO images).  PH = 2: dQ (a wave owns 32 queries, Q_i / dO_i fragments in registers,
// sweeps the key tiles of the K / V images).  The two images of a launch are DOUBLE-BUFFERED (4 x 28 KiB at L = 197): the next
// item's images, row constants and fragments arrive while the current item is computed; the one barrier per item hands the buffers
// over.  Selected by MMK_ATTN_SPLIT=1 in debug-switch builds (measured against the five-product kernel: HISTORY.md 5.2).
template <int NT, int NW, bool DROP, int PH>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_bwd_split_kernel(const AttnBwdArgs a) {
  constexpr int LP = 32 * NT;
  constexpr int IMG = LP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [2 buffers][X | Y] images: PH 1: X = Q, Y = dO;  PH 2: X = K, Y = V
  constexpr int SROWS = NT <= 7 ? 32 : 16;
  float* rowc = reinterpret_cast<float*>(smem + 4 * IMG);  // [2 buffers][lse2 | delta][ROWC]
  char* stage = reinterpret_cast<char*>(rowc + 4 * ROWC) + (threadIdx.x >> 6) * (SROWS * 128);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const ImgLane il = img_lane(lane);
  const long osl = (long)a.H * ATT_DH;
  const float sl2 = a.scale * 1.4426950408889634f;
  const int nitems = a.B * a.H;
  const bool active = wave < NT;  // wave-uniform: this wave owns tile `wave`

  bf16x8 xf[4], yf[4];        // this wave's own fragments of the current item (PH 1: K_j, V_j; PH 2: Q_i, dO_i)
  bf16x8 xn[4], yn[4];        // ... of the next item (second register set: they arrive behind the current item's MFMAs)

  auto obase_of = [&](int bh) { return ((long)(bh / a.H) * a.L * a.H + (bh % a.H)) * ATT_DH; };
  auto load_regs = [&](int bh, bf16x8 (&xo)[4], bf16x8 (&yo)[4]) {
    const int b = bh / a.H, hh = bh % a.H;
    if (active) {
      const int lo = opaque(lane);
      const int jc = min(wave * 32 + (lo & 31), a.L - 1);
      const char* xbase;
      const char* ybase;
      uint32_t xoff, yoff;
      if (PH == 1) {
        xbase = reinterpret_cast<const char*>(a.k + b * a.k_sb + hh * a.k_sh);
        ybase = reinterpret_cast<const char*>(a.v + b * a.v_sb + hh * a.v_sh);
        xoff = (uint32_t)(jc * (int)a.k_sl + 8 * (lo >> 5)) * 2u;
        yoff = (uint32_t)(jc * (int)a.v_sl + 8 * (lo >> 5)) * 2u;
      } else {
        xbase = reinterpret_cast<const char*>(a.q + b * a.q_sb + hh * a.q_sh);
        ybase = reinterpret_cast<const char*>(a.dout + obase_of(bh));
        xoff = (uint32_t)(jc * (int)a.q_sl + 8 * (lo >> 5)) * 2u;
        yoff = (uint32_t)(jc * (int)osl + 8 * (lo >> 5)) * 2u;
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        xo[kk] = *reinterpret_cast<const bf16x8*>(xbase + xoff + 32 * kk);
        yo[kk] = *reinterpret_cast<const bf16x8*>(ybase + yoff + 32 * kk);
      }
    }
    asm volatile("" ::: "memory");  // keep the loads here (not sunk to their first use)
  };
  auto issue_imgs = [&](int bh, int buf) {  // the item's two images + its row-constant record (two pieces, waves 0 and 1)
    const int b = bh / a.H, hh = bh % a.H;
    const int lo = opaque(lane);
    char* X = smem + buf * 2 * IMG;
    char* Y = X + IMG;
    if (PH == 1) {
      img_load(X, a.q + b * a.q_sb + hh * a.q_sh, a.q_sl, a.L, LP, wave, NW, lo);
      img_load(Y, a.dout + obase_of(bh), osl, a.L, LP, wave, NW, lo);
    } else {
      img_load(X, a.k + b * a.k_sb + hh * a.k_sh, a.k_sl, a.L, LP, wave, NW, lo);
      img_load(Y, a.v + b * a.v_sb + hh * a.v_sh, a.v_sl, a.L, LP, wave, NW, lo);
    }
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    if (wave_s < 2)
      lds_dma16(a.delta + ((long)bh * 2 + wave_s) * ROWC, (uint32_t)lo * 16u,
                lds_addr_of(reinterpret_cast<const char*>(rowc + (buf * 2 + wave_s) * ROWC)));
  };

  int item = blockIdx.x;
  if (item < nitems) {
    issue_imgs(item, 0);
    load_regs(item, xf, yf);
    wait_vmem_all();
  }
  for (int n = 0; item < nitems; item += gridDim.x, ++n) {
    const int buf = n & 1;
    const char* X = smem + buf * 2 * IMG;
    const char* Y = X + IMG;
    const float* lse2s = rowc + buf * 2 * ROWC;
    const float* dls = lse2s + ROWC;
    const long gbase = (long)(item / a.H) * a.g_sb + (long)(item % a.H) * ATT_DH;
    const uint32_t dkey = DROP ? drop_key(a.seed_lo, a.seed_hi, (uint32_t)item) : 0u;
    __syncthreads();  // buffer `buf` has landed (every wave waited for its own pieces); every wave is done with the other buffer
    const int next = item + gridDim.x;
    if (next < nitems) {
      issue_imgs(next, buf ^ 1);
      load_regs(next, xn, yn);
    }
    f32x16 acc1[2], acc2[2];
    if (active && PH == 1) {
      // ---------------- dK, dV of key tile `wave` (the key on the lane)
      const int j = wave * 32 + r;
      const bool jvalid = j < a.L;
      f32x16 (&dkt)[2] = acc1;
      f32x16 (&dvt)[2] = acc2;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) dkt[dt][e] = dvt[dt][e] = 0.f;
#pragma unroll 1
      for (int it = 0; it < NT; ++it) {
        f32x16 sc, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[e] = dp[e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(X + it * 4096 + il.row[kk]), xf[kk], sc, 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(Y + it * 4096 + il.row[kk]), yf[kk], dp, 0, 0, 0);
        bf16x8 pf[2], df[2];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float4 l2 = *reinterpret_cast<const float4*>(lse2s + it * 32 + 8 * g4 + 4 * h);
          const float4 dl = *reinterpret_cast<const float4*>(dls + it * 32 + 8 * g4 + 4 * h);
          const float l2v[4] = {l2.x, l2.y, l2.z, l2.w};
          const float dlv[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            const int e = 4 * g4 + e4;
            const float p = jvalid ? att_exp2(fmaf(sc[e], sl2, -l2v[e4])) : 0.f;
            float keep = 1.f;
            if (DROP) {
              const uint32_t w = drop_word(dkey, it * 32 + 8 * g4 + 4 * h + e4, j >> 1);
              keep = ((w >> (16 * (j & 1))) & 0xFFFFu) < a.drop_thr ? 0.f : a.drop_scale;
            }
            pf[e >> 3][e & 7] = (bf16_t)(DROP ? p * keep : p);
            df[e >> 3][e & 7] = (bf16_t)(p * ((DROP ? dp[e] * keep : dp[e]) - dlv[e4]));
          }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(Y, il, it * 4096 + s * 2048, dt), pf[s], dvt[dt], 0, 0, 0);
            dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(X, il, it * 4096 + s * 2048, dt), df[s], dkt[dt], 0, 0, 0);
          }
      }
    }
    if (active && PH == 2) {
      // ---------------- dQ of query tile `wave` (the query on the lane)
      const int i = wave * 32 + r;
      const float l2 = lse2s[i], dl = dls[i];
      f32x16 (&dqt)[2] = acc1;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) dqt[dt][e] = 0.f;
#pragma unroll 1
      for (int jt = 0; jt < NT; ++jt) {
        f32x16 sc, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[e] = dp[e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(X + jt * 4096 + il.row[kk]), xf[kk], sc, 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(Y + jt * 4096 + il.row[kk]), yf[kk], dp, 0, 0, 0);
        if (DROP) {
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            const uint32_t w = drop_word(dkey, i, (jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) >> 1);
            dp[e] *= (w & 0xFFFFu) < a.drop_thr ? 0.f : a.drop_scale;
            dp[e + 1] *= (w >> 16) < a.drop_thr ? 0.f : a.drop_scale;
          }
        }
        bf16x8 df[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) df[e >> 3][e & 7] = (bf16_t)(att_exp2(fmaf(sc[e], sl2, -l2)) * (dp[e] - dl));
        if (jt == NT - 1 && a.L < LP) {  // keys beyond L (finite filler rows): dS = 0
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (jt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= a.L) df[e >> 3][e & 7] = (bf16_t)0.f;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
            dqt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(X, il, jt * 4096 + s * 2048, dt), df[s], dqt[dt], 0, 0, 0);
      }
    }
    wait_vmem_all();  // the next item's pieces and fragments (issued before this item's MFMAs)
    if (active) {
      if (PH == 1) {
        store_rows_staged<SROWS>(stage, a.dk + gbase, a.g_sl, wave * 32, a.L, acc1, a.scale, opaque(lane));
        store_rows_staged<SROWS>(stage, a.dv + gbase, a.g_sl, wave * 32, a.L, acc2, 1.f, opaque(lane));
      } else {
        store_rows_staged<SROWS>(stage, a.dq + gbase, a.g_sl, wave * 32, a.L, acc1, a.scale, opaque(lane));
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        xf[kk] = xn[kk];
        yf[kk] = yn[kk];
      }
    }
  }
}

#endif  // MMK_DEBUG_SWITCHES

// delta_i = sum_d dO[i][d] O[i][d] of query tile t, by ONE wave: lane (r, h) loads columns 32 h .. + 31 of row 32 t + r of O and dO
// (four 16-byte pieces each, straight from global memory: dO is on its way into LDS anyway, so these hit the L2), the two halves
// meet through one cross-lane exchange.  Rows beyond L read row L - 1 and store 0.
__device__ __forceinline__ void delta_load(const AttnBwdArgs& a, long obase, long osl, int t, int lane, bf16x8 (&o4)[4], bf16x8 (&g4)[4]) {
  const int row = min(t * 32 + (lane & 31), a.L - 1);
  const long off = obase + (long)row * osl + 32 * (lane >> 5);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    o4[c] = *reinterpret_cast<const bf16x8*>(a.o + off + 8 * c);
    g4[c] = *reinterpret_cast<const bf16x8*>(a.dout + off + 8 * c);
  }
}
__device__ __forceinline__ void delta_store(const AttnBwdArgs& a, float* dls, int t, int lane, const bf16x8 (&o4)[4], const bf16x8 (&g4)[4]) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      s0 = fmaf((float)o4[c][e], (float)g4[c][e], s0);
      s1 = fmaf((float)o4[c][e + 1], (float)g4[c][e + 1], s1);
    }
  float s = s0 + s1;
  s += __shfl_xor(s, 32);
  const int i = t * 32 + (lane & 31);
  if (lane < 32) dls[i] = i < a.L ? s : 0.f;
}

// ---- five-product backward for NT < NW (a spare wave exists; L = 197 -> 7 key waves + 1, L = 77 -> 3 + 1)
// Waves 0..NT-1 each own 32 KEYS for the whole item and sweep the query tiles in lockstep: S = Q Kᵀ and dP = dO Vᵀ
// with the key on the lane, P and dS = P∘(dP - δ) feed dVᵀ += dOᵀ P and dKᵀ += Qᵀ dS as B operands (as phase 1
// above).  dQ needs the sum over ALL keys, i.e. over the key waves: every key wave also drops its dS tile, transposed,
// into a [key][query] LDS image (4 ds_write_b64 per lane), and after the step's barrier wave NT alone forms
// dQᵀ[it] = Kᵀ dSᵀ over the 32·NT keys (both operands by transposed reads: K image, dS image) while the key waves are
// already on the next query tile (two dS buffers).  Five MFMA products per tile pair instead of seven and no second
// set of exponentials; the dQ wave's 2·NT·2 MFMAs per step balance a key wave's 16 MFMAs + softmax arithmetic.
// STAGED: the key waves are the only loaders (four 8-row pieces of every image each), K first, then the Q / dO pieces in row
// order, and a step only waits for the pieces of ITS query tile (counted vmcnt): step 0 starts when K, V, the row constants and
// the first 32 rows of Q / dO are in, the rest of the 84 KiB lands behind the steps.
constexpr bool bwd5_makes_rowc(int NT, bool STAGED) { return STAGED && NT <= 3; }   // row constants in the kernel (else attn_delta_kernel)

template <int NT, int NW, bool DROP, bool STAGED, bool MASK>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_bwd5_kernel(const AttnBwdArgs a) {
  static_assert(NT < NW, "needs a spare wave for dQ");
  constexpr bool INROWC = bwd5_makes_rowc(NT, STAGED);
  constexpr int LP = 32 * NT;
  constexpr int IMG = LP * 128;
  constexpr int DSB = LP * 64;  // one dS image: [LP keys][32 queries] bf16, 8-byte slot index ^= (key>>1)&7
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* Gs = Qs + IMG;  // dO
  char* Ks = Gs + IMG;
  char* DS = Ks + IMG;                                      // 2 buffers
  constexpr int REC = MASK ? 3 : 2;
  float* rowc = reinterpret_cast<float*>(DS + 2 * DSB);     // [lse2 | delta | (MASK) key bias][ROWC]
  char* stage = reinterpret_cast<char*>(rowc + REC * ROWC) + (threadIdx.x >> 6) * STAGE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const ImgLane il = img_lane(lane);
  const long osl = (long)a.H * ATT_DH;
  const float sl2 = a.scale * 1.4426950408889634f;
  const int nitems = a.B * a.H;
  const bool keyw = wave < NT;   // wave-uniform roles
  const bool dqw = wave == NT;
  const float* lse2s = rowc;
  const float* dls = rowc + ROWC;

  // ---- the dQ wave makes an item's row constants: lse2 = lse log2(e) (+inf beyond L: P = 0 on padded query rows) and
  // delta_i = sum_d dO_id O_id of every query tile (one tile's loads in flight while the previous one is summed).  Plain loads: this
  // wave issues no LDS-DMA, its vmcnt is its own.  It runs AHEAD of the item: for the first item while the key waves' pieces fly,
  // for every later one right after its last dQ tile of the previous item, beside the key waves' final stores and the next load
  // phase (the key waves read the row constants for the last time before the step barrier the dQ wave has just left).  This
  // replaces a streaming pass over O and dO before the kernel (attn_delta_kernel) -- for up to three query tiles (BERT's L = 77:
  // 11 % off the backward, launch of the streaming pass included).  From five tiles on the dQ wave is the pole of the item (its
  // 4 NT MFMAs per step against a key wave's 16 + softmax): the same work there cost what the separate pass costs (L = 197: 1038 vs
  // 1035 us, with four tiles' loads in flight 1042), so those keep attn_delta_kernel and its records.
  auto make_rowc = [&](int it_) {
    const int lo = opaque(lane);
    const long ob = ((long)(it_ / a.H) * a.L * a.H + (it_ % a.H)) * ATT_DH;
    float* rw = rowc;
    const float* lrow = a.lse + (long)it_ * a.L;
    bf16x8 o4[2][4], g4[2][4];
    delta_load(a, ob, osl, 0, lo, o4[0], g4[0]);
#pragma unroll
    for (int k = 0; k < ROWC / 64; ++k) {
      const int l = lo + 64 * k;
      rw[l] = l < a.L ? lrow[min(l, a.L - 1)] * 1.4426950408889634f : INFINITY;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t + 1 < NT) delta_load(a, ob, osl, t + 1, lo, o4[(t + 1) & 1], g4[(t + 1) & 1]);
      delta_store(a, rw + ROWC, t, lo, o4[t & 1], g4[t & 1]);
    }
  };
  int n_mine = 0;
  auto stamp = [&](int k) {
#ifdef MMK_ATTN_STAMPS_BUILD   // debug builds only (make EXTRA=-DMMK_ATTN_STAMPS_BUILD): the checks are not free in this kernel
    if (a.stamps != nullptr && blockIdx.x == 0 && tid == 0 && n_mine < 32) a.stamps[n_mine * 16 + k] = __builtin_readcyclecounter();
#else
    (void)k;
#endif
  };
  for (int item = blockIdx.x; item < nitems; item += gridDim.x, ++n_mine) {
    const int b = item / a.H, hh = item % a.H;
    const long obase = ((long)b * a.L * a.H + hh) * ATT_DH;
    const long gbase = (long)b * a.g_sb + (long)hh * ATT_DH;
    const uint32_t dkey = DROP ? drop_key(a.seed_lo, a.seed_hi, (uint32_t)item) : 0u;
    stamp(0);
    // ---- load: V_j fragments (compiler-visible, first), then the Q, dO, K images and the row-constant record
    bf16x8 kf[4], vf[4];
    if (keyw && !STAGED) {
      const int lo = opaque(lane);
      const uint32_t voff = (uint32_t)(min(wave * 32 + (lo & 31), a.L - 1) * (int)a.v_sl + 8 * (lo >> 5)) * 2u;
      const char* vbase = reinterpret_cast<const char*>(a.v + b * a.v_sb + hh * a.v_sh);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) vf[kk] = *reinterpret_cast<const bf16x8*>(vbase + voff + 32 * kk);
    }
    asm volatile("" ::: "memory");
    // pieces of tile t are complete once every loader has its pieces g <= (4 t + 3) / NT: that leaves 2 (3 - g) younger DMAs
    auto allow = [](int t) { return 2 * (3 - (4 * t + 3) / NT); };
    if (STAGED) {
      const int lo = opaque(lane);
      const int wave_s = __builtin_amdgcn_readfirstlane(wave);
      if (wave_s < NT) {
        // this wave's 32 V rows -> its own staging tile (free until the item's final stores), read back as row fragments below.
        // Every load of the item is an LDS-DMA the compiler does not track: a compiler-visible load would make it wait for ALL
        // of them (vmcnt(0)) at its first use.
#pragma unroll
        for (int g = 0; g < 4; ++g) img_load_piece(stage, a.v + b * a.v_sb + hh * a.v_sh, a.v_sl, a.L, 4 * wave_s + g, lo, g);
#pragma unroll
        for (int g = 0; g < 4; ++g) img_load_piece(Ks, a.k + b * a.k_sb + hh * a.k_sh, a.k_sl, a.L, wave_s + NT * g, lo);
        // row constants: the lse2 / delta records of attn_delta_kernel, or (INROWC) made by the dQ wave; the key bias record of a
        // masked call is one more piece.  All of them are older than the Q / dO pieces, whose counted waits they do not change.
        if (!INROWC) {
          for (int rc = wave_s; rc < 2; rc += NT)
            lds_dma16(a.delta + ((long)item * 2 + rc) * ROWC, (uint32_t)lo * 16u, lds_addr_of(reinterpret_cast<const char*>(rowc + rc * ROWC)));
        }
        if (MASK && wave_s == (NT > 2 ? 2 : 0))
          lds_dma16(a.kbias + (long)b * ROWC, (uint32_t)lo * 16u, lds_addr_of(reinterpret_cast<const char*>(rowc + 2 * ROWC)));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          img_load_piece(Qs, a.q + b * a.q_sb + hh * a.q_sh, a.q_sl, a.L, wave_s + NT * g, lo);
          img_load_piece(Gs, a.dout + obase, osl, a.L, wave_s + NT * g, lo);
        }
      }
      stamp(1);
      if (wave_s < NT) wait_vmem_upto(allow(0));
      if (INROWC && wave_s == NT && n_mine == 0) make_rowc(item);   // (later items: made by the dQ wave at the end of the previous item)
    } else {
      const int lo = opaque(lane);
      img_load(Qs, a.q + b * a.q_sb + hh * a.q_sh, a.q_sl, a.L, LP, wave, NW, lo);
      img_load(Gs, a.dout + obase, osl, a.L, LP, wave, NW, lo);
      img_load(Ks, a.k + b * a.k_sb + hh * a.k_sh, a.k_sl, a.L, LP, wave, NW, lo);
      const int wave_s = __builtin_amdgcn_readfirstlane(wave);
      if (wave_s < 2)
        lds_dma16(a.delta + ((long)item * 2 + wave_s) * ROWC, (uint32_t)lo * 16u,
                  lds_addr_of(reinterpret_cast<const char*>(rowc + wave_s * ROWC)));
      if (MASK && wave_s == 2)
        lds_dma16(a.kbias + (long)b * ROWC, (uint32_t)lo * 16u, lds_addr_of(reinterpret_cast<const char*>(rowc + 2 * ROWC)));
      stamp(1);
      wait_vmem_all();
    }
    __syncthreads();
    stamp(2);

    f32x16 acc1[2], acc2[2];  // key waves: dKᵀ, dVᵀ;  dQ wave: dQᵀ of one query tile (acc1)
    const int j = wave * 32 + r;  // key waves: this lane's key
    const bool jvalid = j < a.L;
    const float kbj = (MASK && keyw) ? rowc[2 * ROWC + j] : 0.f;
    if (keyw) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) kf[kk] = lds_row_frag(Ks + wave * 4096 + il.row[kk]);
      if (STAGED) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) vf[kk] = lds_row_frag(stage + il.row[kk]);
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[dt][e] = acc2[dt][e] = 0.f;
    }
    // per-lane offsets of the dQ wave's transposed reads (natural k order: element jj <-> key 16ks + 8h + jj)
    int ktr[2][2], dtr[2];
    {
      const int li = lane & 15, q = li >> 2, p = li & 3, g1 = (lane >> 4) & 1;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int row = 8 * h + 4 * u + q;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) ktr[u][mt] = row * 128 + (((4 * mt + 2 * g1 + (p >> 1)) ^ img_swz(row)) << 4) + 8 * (p & 1);
        dtr[u] = row * 64 + (((4 * g1 + p) ^ ((row >> 1) & 7)) << 3);
      }
    }

    // The roles run their own loops (the same number of barriers each): their register sets do not overlap, so the dQ wave can
    // keep the transposed K fragments of the whole item -- the same for every query tile -- in registers (28 x 4 VGPRs at NT = 7)
    // instead of re-reading them from LDS at every step (two thirds of its LDS reads, and the reads its MFMAs waited for).
    typedef short s4 __attribute__((ext_vector_type(4)));
    typedef short s8 __attribute__((ext_vector_type(8)));
    auto tr8 = [&](const char* p0, const char* p1) {
      const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p0));
      const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p1));
      s8 f;
      f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
      f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
      return __builtin_bit_cast(bf16x8, f);
    };
    if (keyw) {
    // (r03: issuing the S / dP products of tile it + 1 ahead of the softmax arithmetic of tile it -- a software pipeline over the
    // query tiles, +32 VGPRs -- saved cycles in the stamps but no time: 1045-1060 us against 1053-1056 at L = 197; not kept)
#pragma unroll 1
    for (int it = 0; it <= NT; ++it) {
      if (it < NT) {
        f32x16 sc, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[e] = dp[e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(Qs + it * 4096 + il.row[kk]), kf[kk], sc, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_row_frag(Gs + it * 4096 + il.row[kk]), vf[kk], dp, 0, 0, 0);
        }
        // ---------------- pair (query tile it, key tile wave), the key on the lane
        if (it == 3) stamp(12);
        bf16x8 pf[2], df[2];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float4 l2 = *reinterpret_cast<const float4*>(lse2s + it * 32 + 8 * g4 + 4 * h);
          const float4 dl = *reinterpret_cast<const float4*>(dls + it * 32 + 8 * g4 + 4 * h);
          const float l2v[4] = {l2.x, l2.y, l2.z, l2.w};
          const float dlv[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            const int e = 4 * g4 + e4;
            const float p = bwd_prob<MASK>(sc[e], sl2, kbj, l2v[e4], jvalid, a.causal, j, it * 32 + 8 * g4 + 4 * h + e4);
            float keep = 1.f;
            if (DROP) {
              const uint32_t w = drop_word(dkey, it * 32 + 8 * g4 + 4 * h + e4, j >> 1);
              keep = ((w >> (16 * (j & 1))) & 0xFFFFu) < a.drop_thr ? 0.f : a.drop_scale;
            }
            pf[e >> 3][e & 7] = (bf16_t)(DROP ? p * keep : p);
            df[e >> 3][e & 7] = (bf16_t)(p * ((DROP ? dp[e] * keep : dp[e]) - dlv[e4]));
          }
        }
        if (it == 3) stamp(13);
        // dS tile -> [key][query] image of this step: registers 4g..4g+3 = queries 8g + 4h .. +3 of key j
        {
          char* dsrow = DS + (it & 1) * DSB + j * 64;
          const int sw = (j >> 1) & 7;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            bf16x4 w4;
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) w4[e4] = df[g4 >> 1][4 * (g4 & 1) + e4];
            *reinterpret_cast<bf16x4*>(dsrow + (((2 * g4 + h) ^ sw) << 3)) = w4;
          }
        }
        if (it == 3) stamp(14);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            acc2[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(Gs, il, it * 4096 + s * 2048, dt), pf[s], acc2[dt], 0, 0, 0);
            acc1[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_tr_frag(Qs, il, it * 4096 + s * 2048, dt), df[s], acc1[dt], 0, 0, 0);
          }
        if (it == 3) stamp(15);
      }
      if (STAGED && it + 1 < NT) wait_vmem_upto(allow(it + 1));   // this wave's pieces of the next query tile
      __syncthreads();
      stamp(3 + it);
    }
    } else if (dqw) {
    constexpr int KH = 2 * NT < 12 ? 2 * NT : 12;   // k steps whose K fragments stay in registers (all but the last two at NT = 7: no spills)
    bf16x8 kt[KH][2];
#pragma unroll
    for (int ks = 0; ks < KH; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) kt[ks][mt] = tr8(Ks + ks * 2048 + ktr[0][mt], Ks + ks * 2048 + ktr[1][mt]);
#pragma unroll 1
    for (int it = 0; it <= NT; ++it) {
      if (it > 0) {
        // ---------------- dQᵀ of query tile it-1 = Kᵀ dSᵀ over all keys
        const char* dsb = DS + ((it - 1) & 1) * DSB;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc1[dt][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2 * NT; ++ks) {
          const bf16x8 bfr = tr8(dsb + ks * 1024 + dtr[0], dsb + ks * 1024 + dtr[1]);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ks < KH ? kt[ks < KH ? ks : 0][mt] : tr8(Ks + ks * 2048 + ktr[0][mt], Ks + ks * 2048 + ktr[1][mt]),
                                                               bfr, acc1[mt], 0, 0, 0);
        }
        float csq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        store_rows_staged_cs(stage, a.dq + gbase, a.g_sl, (it - 1) * 32, a.L, acc1, a.scale, opaque(lane), csq);
        cs_flush(stage, csq, lane, cs_slot(a, b, hh, it - 1, 0));
      }
      __syncthreads();
    }
    if (INROWC && item + (int)gridDim.x < nitems) make_rowc(item + (int)gridDim.x);
    } else {
#pragma unroll 1
      for (int it = 0; it <= NT; ++it) __syncthreads();
    }
    if (keyw) {
      {
        float csk[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        store_rows_staged_cs(stage, a.dk + gbase, a.g_sl, wave * 32, a.L, acc1, a.scale, opaque(lane), csk);
        cs_flush(stage, csk, lane, cs_slot(a, b, hh, wave, 1));
      }
      {
        float csv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        store_rows_staged_cs(stage, a.dv + gbase, a.g_sl, wave * 32, a.L, acc2, 1.f, opaque(lane), csv);
        cs_flush(stage, csv, lane, cs_slot(a, b, hh, wave, 2));
      }
    }
    stamp(4 + NT);
  }
}

template <int NT, int NW, bool DROP, bool STAGED, bool MASK>
static int launch_attn_bwd5(const AttnBwdArgs& a, hipStream_t st) {
  constexpr int LP = 32 * NT;
  constexpr int bytes = 3 * LP * 128 + 2 * LP * 64 + (MASK ? 3 : 2) * ROWC * 4 + NW * STAGE_BYTES;
  static_assert(bytes <= 160 * 1024, "LDS budget");
  auto kern = attn_bwd5_kernel<NT, NW, DROP, STAGED, MASK>;
  KernelSetup ks;   // LDS opt-in, occupancy and CU count of this kernel on the current device
  if (int rc = kernel_setup(reinterpret_cast<const void*>(kern), 64 * NW, bytes, &ks)) return rc;
  const int cus = ks.cus, wgs_per_cu = ks.wgs_per_cu;
  const int grid = std::min(a.B * a.H, cus * wgs_per_cu * att_grid_factor());
  if (!bwd5_makes_rowc(NT, STAGED)) {   // short sequences make lse2 / delta in the kernel (its dQ wave); the others read this pass's records
    const long chunks = (long)a.B * a.L * a.H * 8;
    const int dgrid = (int)std::min<long>((chunks + 255) / 256, (long)cus * 16);
    hipLaunchKernelGGL(attn_delta_kernel, dim3(dgrid), dim3(256), 0, st, a.o, a.dout, a.lse, a.delta, a.B, a.H, a.L);
  }
  ProfEvents pe(MMK_K_ATTN_BWD);
  hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), bytes, st, pe.start, pe.stop, 0, a);
  MMK_LAUNCH_CHECK();
  return 0;
}

template <int NT, int NW, bool DROP, bool MASK>
static int launch_attn_bwd(const AttnBwdArgs& a, hipStream_t st) {
  constexpr int LP = 32 * NT;
  constexpr int bytes = 4 * LP * 128 + 2 * (MASK ? 3 : 2) * ROWC * 4 + NW * (NT <= 7 ? 32 : 16) * 128;
  static_assert(bytes <= 160 * 1024, "LDS budget");
  auto kern = attn_bwd_kernel<NT, NW, DROP, MASK>;
  KernelSetup ks;   // LDS opt-in, occupancy and CU count of this kernel on the current device
  if (int rc = kernel_setup(reinterpret_cast<const void*>(kern), 64 * NW, bytes, &ks)) return rc;
  const int cus = ks.cus, wgs_per_cu = ks.wgs_per_cu;
  const int grid = std::min(a.B * a.H, cus * wgs_per_cu * att_grid_factor());
  {
    const long chunks = (long)a.B * a.L * a.H * 8;
    const int dgrid = (int)std::min<long>((chunks + 255) / 256, (long)cus * 16);
    hipLaunchKernelGGL(attn_delta_kernel, dim3(dgrid), dim3(256), 0, st, a.o, a.dout, a.lse, a.delta, a.B, a.H, a.L);
  }
  ProfEvents pe(MMK_K_ATTN_BWD);
  hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), bytes, st, pe.start, pe.stop, 0, a);
  MMK_LAUNCH_CHECK();
  return 0;
}

#ifdef MMK_DEBUG_SWITCHES
template <int NT, int NW, bool DROP>
static int launch_attn_bwd_split(const AttnBwdArgs& a, hipStream_t st) {
  constexpr int LP = 32 * NT;
  constexpr int bytes = 4 * LP * 128 + 4 * ROWC * 4 + NW * (NT <= 7 ? 32 : 16) * 128;
  static_assert(bytes <= 160 * 1024, "LDS budget");
  auto k1 = attn_bwd_split_kernel<NT, NW, DROP, 1>;
  auto k2 = attn_bwd_split_kernel<NT, NW, DROP, 2>;
  KernelSetup ks, ks2;   // both kernels of the split form: LDS opt-in per device; the grid follows the first one's occupancy
  if (int rc = kernel_setup(reinterpret_cast<const void*>(k1), 64 * NW, bytes, &ks)) return rc;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(k2), 64 * NW, bytes, &ks2)) return rc;
  const int cus = ks.cus, wgs_per_cu = ks.wgs_per_cu;
  const int grid = std::min(a.B * a.H, cus * wgs_per_cu * att_grid_factor());
  {
    const long chunks = (long)a.B * a.L * a.H * 8;
    const int dgrid = (int)std::min<long>((chunks + 255) / 256, (long)cus * 16);
    hipLaunchKernelGGL(attn_delta_kernel, dim3(dgrid), dim3(256), 0, st, a.o, a.dout, a.lse, a.delta, a.B, a.H, a.L);
  }
  {
    ProfEvents pe(MMK_K_ATTN_BWD);
    hipExtLaunchKernelGGL(k1, dim3(grid), dim3(64 * NW), bytes, st, pe.start, pe.stop, 0, a);
  }
  {
    ProfEvents pe(MMK_K_ATTN_BWD);
    hipExtLaunchKernelGGL(k2, dim3(grid), dim3(64 * NW), bytes, st, pe.start, pe.stop, 0, a);
  }
  MMK_LAUNCH_CHECK();
  return 0;
}
#endif  // MMK_DEBUG_SWITCHES

}  // namespace mmk

using namespace mmk;

namespace {
#define MMK_ATTN_STRIDES_OK(a)                                                                                       \
  ((a.q_sl % 8 == 0) && (a.k_sl % 8 == 0) && (a.v_sl % 8 == 0) && (a.q_sh % 8 == 0) && (a.k_sh % 8 == 0) &&        \
   (a.v_sh % 8 == 0) && (a.q_sb % 8 == 0) && (a.k_sb % 8 == 0) && (a.v_sb % 8 == 0))
}  // namespace

// ---- key bias records: one [ROWC] row per sample from a padding mask in any of the forms the callers hold
namespace mmk {
template <typename T, bool ADDITIVE>
__global__ __launch_bounds__(256) void attn_key_bias_kernel(const T* __restrict__ mask, long sb, const int* __restrict__ lengths, int B, int L,
                                                            float* __restrict__ rec) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * ROWC) return;
  const int b = (int)(idx / ROWC), l = (int)(idx % ROWC);
  float v = -INFINITY;
  if (l < L) {
    if (lengths) {
      v = l < lengths[b] ? 0.f : KB_MASKED;
    } else if (ADDITIVE) {   // an additive mask in natural-log units (0 / finfo.min, or any finite bias); NaN counts as masked
      const float x = (float)mask[b * sb + l] * 1.4426950408889634f;
      v = x > KB_MASKED ? x : KB_MASKED;
    } else {
      v = mask[b * sb + l] != (T)0 ? 0.f : KB_MASKED;
    }
  }
  rec[idx] = v;
}
}  // namespace mmk

extern "C" int mmk_attn_key_bias(const void* mask, int kind, int B, int L, int64_t mask_sb, float* rec, void* stream) {
  MMK_REQUIRE(mask && rec, "attn_key_bias: null pointer");
  MMK_REQUIRE(B > 0 && L > 0 && L <= ROWC, "attn_key_bias: need 1 <= L <= 256");
  MMK_REQUIRE(kind == MMK_KEYMASK_LENGTHS || mask_sb >= L, "attn_key_bias: mask rows overlap");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(((long)B * ROWC + 255) / 256)), block(256);
  switch (kind) {
    case MMK_KEYMASK_LENGTHS:
      hipLaunchKernelGGL((attn_key_bias_kernel<uint8_t, false>), grid, block, 0, st, nullptr, 0l, static_cast<const int*>(mask), B, L, rec);
      break;
    case MMK_KEYMASK_U8:
      hipLaunchKernelGGL((attn_key_bias_kernel<uint8_t, false>), grid, block, 0, st, static_cast<const uint8_t*>(mask), (long)mask_sb, nullptr, B, L, rec);
      break;
    case MMK_KEYMASK_I32:
      hipLaunchKernelGGL((attn_key_bias_kernel<int32_t, false>), grid, block, 0, st, static_cast<const int32_t*>(mask), (long)mask_sb, nullptr, B, L, rec);
      break;
    case MMK_KEYMASK_I64:
      hipLaunchKernelGGL((attn_key_bias_kernel<int64_t, false>), grid, block, 0, st, static_cast<const int64_t*>(mask), (long)mask_sb, nullptr, B, L, rec);
      break;
    case MMK_KEYMASK_F32_KEEP:
      hipLaunchKernelGGL((attn_key_bias_kernel<float, false>), grid, block, 0, st, static_cast<const float*>(mask), (long)mask_sb, nullptr, B, L, rec);
      break;
    case MMK_KEYMASK_F32_ADD:
      hipLaunchKernelGGL((attn_key_bias_kernel<float, true>), grid, block, 0, st, static_cast<const float*>(mask), (long)mask_sb, nullptr, B, L, rec);
      break;
    case MMK_KEYMASK_BF16_ADD:
      hipLaunchKernelGGL((attn_key_bias_kernel<bf16_t, true>), grid, block, 0, st, static_cast<const bf16_t*>(mask), (long)mask_sb, nullptr, B, L, rec);
      break;
    default:
      MMK_REQUIRE(false, "attn_key_bias: unknown mask kind");
  }
  MMK_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmk_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int B, int H, int L, int dh,
                            const int64_t* q_strides, const int64_t* k_strides, const int64_t* v_strides, float scale,
                            float dropout_p, uint64_t seed, const float* key_bias, int causal, void* stream) {
  MMK_REQUIRE(q && k && v && out && lse && q_strides && k_strides && v_strides, "null pointer");
  MMK_REQUIRE(B > 0 && H > 0 && L > 0, "empty problem");
  MMK_REQUIRE(dh == ATT_DH, "attention kernel supports head_dim 64");
  MMK_REQUIRE(L <= 256, "attention kernel supports sequence length <= 256");
  MMK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0, 1)");
  MMK_REQUIRE(scale > 0.f, "scale must be positive");
  AttnArgs a;
  a.q = static_cast<const bf16_t*>(q); a.k = static_cast<const bf16_t*>(k); a.v = static_cast<const bf16_t*>(v);
  a.out = static_cast<bf16_t*>(out); a.lse = lse;
  a.q_sb = q_strides[0]; a.q_sh = q_strides[1]; a.q_sl = q_strides[2];
  a.k_sb = k_strides[0]; a.k_sh = k_strides[1]; a.k_sl = k_strides[2];
  a.v_sb = v_strides[0]; a.v_sh = v_strides[1]; a.v_sl = v_strides[2];
  MMK_REQUIRE(MMK_ATTN_STRIDES_OK(a), "q/k/v rows must be 16-byte aligned");
  a.B = B; a.H = H; a.L = L; a.scale = scale;
  a.kbias = key_bias; a.causal = causal ? 1 : 0;
  MMK_REQUIRE(!causal || key_bias, "attn_fwd: a causal call needs a key bias record (all-zero for no padding)");
  const bool drop = drop_params(dropout_p, seed, &a.seed_lo, &a.seed_hi, &a.drop_thr, &a.drop_scale);
  const bool mask = key_bias != nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define MMK_ATTN_FWD_CASE(NT)                                                                                 \
  case NT:                                                                                                    \
    if (mask) return drop ? launch_attn_fwd<NT, true, true>(a, st) : launch_attn_fwd<NT, false, true>(a, st); \
    return drop ? launch_attn_fwd<NT, true, false>(a, st) : launch_attn_fwd<NT, false, false>(a, st);
  switch ((L + 31) / 32) {
    MMK_ATTN_FWD_CASE(1) MMK_ATTN_FWD_CASE(2) MMK_ATTN_FWD_CASE(3) MMK_ATTN_FWD_CASE(4)
    MMK_ATTN_FWD_CASE(5) MMK_ATTN_FWD_CASE(6) MMK_ATTN_FWD_CASE(7)
    default: MMK_ATTN_FWD_CASE(8)
  }
}

// debug builds, MMK_ATTN_STAMPS=1: a 4-KiB device buffer that workgroup 0 of the five-product backward stamps (see AttnBwdArgs.stamps)
static unsigned long long* attn_stamp_buffer() {
  static unsigned long long* buf = nullptr;
  static bool tried = false;
  if (!tried) {
    tried = true;
    if (MMK_DBG_ENV("MMK_ATTN_STAMPS") != nullptr && hipMalloc(reinterpret_cast<void**>(&buf), 32 * 16 * sizeof(unsigned long long)) != hipSuccess) buf = nullptr;
  }
  return buf;
}
extern "C" int mmk_attn_debug_stamps(unsigned long long* out, int n) {
  unsigned long long* buf = attn_stamp_buffer();
  MMK_REQUIRE(buf != nullptr && out != nullptr && n > 0 && n <= 32 * 16, "no stamp buffer (needs a debug-switch + attention-stamps build of the library, see tools/attn_bwd_phases.py)");
  MMK_HIP(hipDeviceSynchronize());
  MMK_HIP(hipMemcpy(out, buf, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
  return 0;
}

// 1 when mmk_attn_bwd can fill colsum_part for sequences of L rows (the five-product kernel serves them: every tile count
// with a spare wave, i.e. all but 97..128 and 225..256 rows), 0 otherwise.
extern "C" int mmk_attn_bwd_has_colsum(int L) {
  static const bool seven = MMK_DBG_ENV("MMK_ATTN_BWD7") != nullptr;
  const int nt = (L + 31) / 32;
  return !seven && L > 0 && nt != 4 && nt < 8;
}

extern "C" int mmk_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                            float* delta_ws, void* dq, void* dk, void* dv, int B, int H, int L, int dh, const int64_t* q_strides,
                            const int64_t* k_strides, const int64_t* v_strides, const int64_t* grad_strides, float scale,
                            float dropout_p, uint64_t seed, float* colsum_part, const float* key_bias, int causal, void* stream) {
  MMK_REQUIRE(q && k && v && out && dout && lse && delta_ws && dq && dk && dv && q_strides && k_strides && v_strides &&
                  grad_strides,
              "null pointer");
  MMK_REQUIRE(B > 0 && H > 0 && L > 0, "empty problem");
  MMK_REQUIRE(dh == ATT_DH, "attention kernel supports head_dim 64");
  MMK_REQUIRE(L <= 256, "attention kernel supports sequence length <= 256");
  MMK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_p must be in [0, 1)");
  AttnBwdArgs a;
  a.q = static_cast<const bf16_t*>(q); a.k = static_cast<const bf16_t*>(k); a.v = static_cast<const bf16_t*>(v);
  a.o = static_cast<const bf16_t*>(out); a.dout = static_cast<const bf16_t*>(dout); a.lse = lse; a.delta = delta_ws;
  a.dq = static_cast<bf16_t*>(dq); a.dk = static_cast<bf16_t*>(dk); a.dv = static_cast<bf16_t*>(dv);
  a.cs = colsum_part;
  a.stamps = attn_stamp_buffer();
  a.g_sb = grad_strides[0]; a.g_sl = grad_strides[1];
  MMK_REQUIRE(a.g_sb % 8 == 0 && a.g_sl % 8 == 0 && a.g_sl >= (long)H * ATT_DH, "gradient rows must be 16-byte aligned");
  a.q_sb = q_strides[0]; a.q_sh = q_strides[1]; a.q_sl = q_strides[2];
  a.k_sb = k_strides[0]; a.k_sh = k_strides[1]; a.k_sl = k_strides[2];
  a.v_sb = v_strides[0]; a.v_sh = v_strides[1]; a.v_sl = v_strides[2];
  MMK_REQUIRE(MMK_ATTN_STRIDES_OK(a), "q/k/v rows must be 16-byte aligned");
  a.B = B; a.H = H; a.L = L; a.scale = scale;
  a.kbias = key_bias; a.causal = causal ? 1 : 0;
  MMK_REQUIRE(!causal || key_bias, "attn_bwd: a causal call needs the forward's key bias record");
  const bool drop = drop_params(dropout_p, seed, &a.seed_lo, &a.seed_hi, &a.drop_thr, &a.drop_scale);
  const bool mask = key_bias != nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  static const bool seven = MMK_DBG_ENV("MMK_ATTN_BWD7") != nullptr;  // force the seven-product kernel (A/B runs)
#ifdef MMK_DEBUG_SWITCHES
  const bool split = MMK_DBG_ENV("MMK_ATTN_SPLIT") != nullptr && atoi(MMK_DBG_ENV("MMK_ATTN_SPLIT")) != 0;   // two-kernel form (A/B, read per call)
#endif
  static const bool staged = MMK_DBG_ENV("MMK_ATTN_STAGED") ? atoi(MMK_DBG_ENV("MMK_ATTN_STAGED")) != 0 : kAttnStagedDefault;
  MMK_REQUIRE(!colsum_part || mmk_attn_bwd_has_colsum(L), "attn_bwd: column sums are not available for this sequence length");
#define MMK_ATTN_BWD7(NT, NW)                                                                                          \
  do {                                                                                                                 \
    if (mask) return drop ? launch_attn_bwd<NT, NW, true, true>(a, st) : launch_attn_bwd<NT, NW, false, true>(a, st);  \
    return drop ? launch_attn_bwd<NT, NW, true, false>(a, st) : launch_attn_bwd<NT, NW, false, false>(a, st);          \
  } while (0)
#define MMK_ATTN_BWD_CASE(NT, NW) \
  case NT: MMK_ATTN_BWD7(NT, NW);
#ifdef MMK_DEBUG_SWITCHES
#define MMK_ATTN_SPLIT_CASE(NT, NW) \
  if (split && !colsum_part && !mask) return drop ? launch_attn_bwd_split<NT, NW, true>(a, st) : launch_attn_bwd_split<NT, NW, false>(a, st);
#else
#define MMK_ATTN_SPLIT_CASE(NT, NW)
#endif
  // masked calls always take the staged five-product form (the unstaged one is an A/B switch of the unmasked kernels)
#define MMK_ATTN_BWD5_CASE(NT, NW)                                                                                                  \
  case NT:                                                                                                                          \
    MMK_ATTN_SPLIT_CASE(NT, NW)                                                                                                     \
    if (seven) MMK_ATTN_BWD7(NT, NW);                                                                                               \
    if (mask) return drop ? launch_attn_bwd5<NT, NW, true, true, true>(a, st) : launch_attn_bwd5<NT, NW, false, true, true>(a, st); \
    if (staged) return drop ? launch_attn_bwd5<NT, NW, true, true, false>(a, st) : launch_attn_bwd5<NT, NW, false, true, false>(a, st); \
    return drop ? launch_attn_bwd5<NT, NW, true, false, false>(a, st) : launch_attn_bwd5<NT, NW, false, false, false>(a, st);
  switch ((L + 31) / 32) {
    MMK_ATTN_BWD5_CASE(1, 4) MMK_ATTN_BWD5_CASE(2, 4) MMK_ATTN_BWD5_CASE(3, 4) MMK_ATTN_BWD_CASE(4, 4)
    MMK_ATTN_BWD5_CASE(5, 8) MMK_ATTN_BWD5_CASE(6, 8) MMK_ATTN_BWD5_CASE(7, 8)
    default: MMK_ATTN_BWD7(8, 8);
  }
}
