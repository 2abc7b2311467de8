// weight-gradient GEMM of the encoders' Linear layers,  dW[N, K] = dY^T x  with  dY [M, N], x [M, K]
// row-major and the contraction over the M = batch * tokens rows (201,728 for ViT-B/16 at B = 1024).  For the
// attention output projections the result is only 768 x 768: nine 256 x 256 tiles cannot fill 256 CUs, and the
// library's answer to that (hipBLASLt, any operand layout, also after TunableOp) runs at 300-380 TFLOP/s -- 0.7 ms for a
// GEMM whose operands stream from HBM in 0.12 ms.  This kernel splits M across the chip instead:
//   * work unit = (M split, 256 x 256 output tile); all tiles of a split are placed on ONE XCD and walk the rows in
//     step, so each dY / x row block comes from HBM once and serves the split's other tiles from that XCD's L2;
//   * both operands have the contraction along their ROWS, so neither can be read as an MFMA fragment directly: the
//     [64 rows][256 columns] stages are filled by LDS-DMA (global_load_lds_dwordx4, swizzled on the source chunk) and
//     read with the hardware transpose read ds_read_b64_tr_b16 -- no transposed copy of dY or x ever exists;
//   * 16 waves as 4 x 4, 64 x 64 per wave = 4 accumulator tiles of v_mfma_f32_32x32x16_bf16 (four waves per SIMD hide
//     the LDS latency without software pipelining), two LDS stages (128 KiB);
//   * f32 partial tiles go to a workspace, a second kernel sums the splits (deterministic, no atomics) and writes dW in
//     the parameter's dtype.
#include <hip/hip_ext.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace mmk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WG_TILE = 256;   // output tile edge of the full-size kernel
constexpr int WG_BM = 64;      // contraction rows per LDS stage
[[maybe_unused]] constexpr int WG_STAGE = WG_BM * WG_TILE * 2;  // bytes of one operand stage (32 KiB) at the 256 tile (debug-switch kernel)
// Narrow layers (HTSAT's 96 / 288 / 384-wide Linears, the I-JEPA predictor's 384: mmlearn/modules/encoders/vision.py:397-569) fill
// a 256 x 256 tile to 14-56 %: the same kernel is instantiated on 128 x 128 tiles -- four waves (2 x 2, still 64 x 64 per wave),
// 16 KiB stages, 64 KiB of LDS so that TWO workgroups share a CU -- and wg_tile_edge() picks it where it saves padded work.
constexpr int WG_TILE_NARROW = 128;
// which kernel serves full 256-tile weights: the 8-phase form (wgrad8_kernel, below) or the two-stage one (A/B: MMK_WGRAD_KERNEL)
constexpr bool kWgrad8Default = true;

// Output tile edge for an [N, K] weight: 128 when the 128-tiles cover it with at most 0.8 x the padded area of the 256-tiles.
static inline int wg_tile_edge(int N, int K) {
  if (const char* e = MMK_DBG_ENV("MMK_WGRAD_TILE")) {   // A/B in debug-switch builds: force one tile edge
    if (atoi(e) == 256) return WG_TILE;
    if (atoi(e) == 128) return WG_TILE_NARROW;
  }
  const long a256 = (long)cdiv(N, 256) * cdiv(K, 256) * 256 * 256;
  const long a128 = (long)cdiv(N, 128) * cdiv(K, 128) * 128 * 128;
  return 5 * a128 <= 4 * a256 ? WG_TILE_NARROW : WG_TILE;
}

struct WgradArgs {
  const bf16_t* dy;  // [M, N], row stride ldy
  const bf16_t* x;   // [M, K], row stride ldx
  float* ws;         // [splits][N_pad][K_pad] partial tiles (N_pad, K_pad = multiples of the tile edge)
  long ldy, ldx;
  int M, N, K;
  int splits, tiles_n, tiles_k, rows_per_split;
  int aligned;       // unit map: 1 = every XCD hosts whole splits (T tiles each), 0 = positions of an XCD run through splits and tiles
  unsigned long long* stamps;   // diagnostic builds (-DMMK_WGRAD_STAMPS_BUILD) only: per wave of workgroup 0, summed segment clocks
};

// Splits of M and the unit map for T tiles on `per_xcd` workgroup slots per XCD (8 XCDs, one round).  XCD-aligned splits keep all
// tiles of a split on one L2; when whole splits leave more than 1/8 of the slots idle (T = 9: 27 of 32, T = 27: 27 of 32, T = 36 on
// the 64-slot narrow kernel: 36 of 64) the positions of an XCD run through (split, tile) pairs instead and the splits fill the chip.
static inline void wg_plan_units(int T, int per_xcd, int* splits, int* aligned) {
  const int s_al = T <= per_xcd ? 8 * (per_xcd / T) : 0;
  const int s_gen = std::max(1, 8 * per_xcd / T);
  bool al = s_al > 0 && (long)s_al * T * 8 >= (long)7 * 8 * per_xcd;
  if (const char* e = MMK_DBG_ENV("MMK_WGRAD_MAP")) {   // A/B in debug-switch builds: 1 = aligned whenever possible (rounds 1-4)
    if (atoi(e) == 1 && s_al > 0) al = true;
  }
  *splits = al ? s_al : s_gen;
  *aligned = al ? 1 : 0;
}

__device__ __forceinline__ void wg_dma16(const void* sbase, uint32_t voff, uint32_t lds_addr) {
  const uint64_t pb = reinterpret_cast<uint64_t>(sbase);
  const uint64_t ps = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pb >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pb);
  lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(reinterpret_cast<const void*>(ps)), "s"(lds_addr) : "memory");
}
// a pointer the compiler must keep in scalar registers (it is wave-uniform by construction; this makes it provably so)
__device__ __forceinline__ const char* wg_uniform(const char* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int wg_opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// LDS stage image: 64 rows x 512 B, 16-byte chunk index (0..31) ^= (row & 3) << 2: the four rows of a transposed-read
// block land in four different 64-byte groups of a 256-byte bank window (conflict-free ds_read_b64_tr_b16).
// Fill: piece p (0..31) = rows 2p, 2p+1; lane l writes chunk position l & 31 of row 2p + (l >> 5).
// TE = 128 (256-byte rows, 16 chunks, same swizzle): one DMA instruction covers FOUR rows, piece p = rows 4p .. 4p + 3, four pieces per
// wave of the four-wave workgroup.
template <int TE>
__device__ __forceinline__ void wg_fill_te(char* stage, const bf16_t* base, long ld, int col0, int ncols, long row0, long row_end, int wave,
                                           int lane) {
  constexpr int CH = TE / 8;                 // 16-byte chunks per row
  constexpr int RP = 64 / CH;                // rows per DMA instruction (1 KiB)
  constexpr int NW = (TE / 64) * (TE / 64);  // waves of the workgroup
  constexpr int PPW = (WG_BM / RP) / NW;     // pieces per wave
  const uint32_t saddr = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)stage);
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = wave_s * PPW + i;
    const int rr = RP * p + lane / CH;
    const int ch = (lane & (CH - 1)) ^ ((rr & 3) << 2);
    const long srow = min(row0 + RP * p, row_end - 1);                 // wave-uniform part of the address
    const int drow = (int)(min(row0 + rr, row_end - 1) - srow);        // 0 .. RP - 1
    const int col = min(col0 + ch * 8, ncols - 8);
    wg_dma16(base + srow * ld, (uint32_t)(drow * (int)ld + col) * 2u, saddr + p * 1024);
  }
}

template <bool SW16 = false>
__device__ __forceinline__ void wg_fill(char* stage, const bf16_t* base, long ld, int col0, int ncols, long row0, long row_end, int wave,
                                        int lane) {
  const uint32_t saddr = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)stage);
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = wave_s * 2 + i;
    const int rr = 2 * p + (lane >> 5);
    // SW16 (16x16x32 MFMA reads): a 32-lane half reads two blocks 8 rows apart in the same columns, so bit 3 of the row also
    // moves the chunk (by 32 bytes)
    const int ch = (lane & 31) ^ ((rr & 3) << 2) ^ (SW16 ? ((rr >> 3) & 1) << 1 : 0);
    // rows past the split's end and columns past the operand's width read a valid address (row / column clamped); the
    // main loop zeroes the fragments of out-of-range rows, out-of-range columns only reach workspace padding
    const long srow = min(row0 + 2 * p, row_end - 1);                  // wave-uniform part of the address
    const int drow = (int)(min(row0 + rr, row_end - 1) - srow);        // 0 or 1
    const int col = min(col0 + ch * 8, ncols - 8);
    wg_dma16(base + srow * ld, (uint32_t)(drow * (int)ld + col) * 2u, saddr + p * 1024);
  }
}

template <typename OUT>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, OUT* __restrict__ dw, long ldw, int N, int K,
                                                           int n_pad, int k_pad, int splits) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int k4 = K / 4;
  if (idx >= (long)N * k4) return;
  const int n = (int)(idx / k4), k = (int)(idx % k4) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < splits; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(ws + ((size_t)s * n_pad + n) * k_pad + k);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  Vec4<OUT>::store(dw + (long)n * ldw + k, acc);
}

// Many splits of a small weight (narrow layers over ~1 M rows: 256-512 splits of a 96 x 96 tile): the loop over splits of the kernel
// above is a few thousand threads each walking hundreds of dependent-latency loads (150 us for 19 MB).  Here 16 lanes share one
// output float4 and stride through the splits, then fold through LDS in a fixed order (deterministic, like the serial sum).
template <typename OUT>
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ ws, OUT* __restrict__ dw, long ldw, int N, int K,
                                                                int n_pad, int k_pad, int splits) {
  __shared__ float4 sm[16][16];
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long idx = (long)blockIdx.x * 16 + o;
  const int k4 = K / 4;
  const bool valid = idx < (long)N * k4;
  const int n = valid ? (int)(idx / k4) : 0, k = valid ? (int)(idx % k4) * 4 : 0;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid)
    for (int s = g; s < splits; s += 16) {
      const float4 v = *reinterpret_cast<const float4*>(ws + ((size_t)s * n_pad + n) * k_pad + k);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  sm[g][o] = acc;
  __syncthreads();
#pragma unroll
  for (int half = 8; half >= 1; half >>= 1) {
    if (g < half) {
      const float4 a = sm[g][o], b = sm[g + half][o];
      sm[g][o] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    __syncthreads();
  }
  if (g == 0 && valid) Vec4<OUT>::store(dw + (long)n * ldw + k, sm[0][o]);
}

#ifdef MMK_DEBUG_SWITCHES
// EXPERIMENT (debug-switch builds only, MMK_WGRAD_MFMA=16; measured, not kept: 1-11 % SLOWER than 32x32x16 on all eight encoder
// shapes in an interleaved same-process A/B, profiles/r03_wgrad_mfma_shape_ab.json -- this kernel is fed by L2 -> LDS, not bound
// by the matrix pipe's clock).  The same kernel on v_mfma_f32_16x16x32_bf16 (/opt/skills/guides/MI355X_MICROARCH.md, DVFS item 7: on random
// data the part holds a higher clock on this shape than on 32x32x16 at equal cycles per FLOP).  A wave still owns 64 x 64 of the
// output: 4 x 4 tiles of 16 x 16 (64 accumulator registers, as before), one 32-row step = 4 + 4 operand fragments (two transposed
// reads each, 16 reads) and 16 MFMAs.  Lane l of a fragment holds rows 8 (l >> 4) .. + 7 of the step for column l & 15: each
// 16-lane group reads its own 4-row block, the two groups of a half 8 rows apart -- hence the extra swizzle bit of the fill.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024, 1) void wgrad_kernel16(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A | B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = a.tiles_n * a.tiles_k;
  int split, tn, tk;
  if (a.aligned) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    split = (slot / T) * 8 + xcd;
    const int tile = slot % T;
    tn = tile / a.tiles_k;
    tk = tile % a.tiles_k;
  } else {
    const int per_xcd = (a.splits * T + 7) >> 3;
    const int pos = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    split = pos / T;
    const int q = pos % T;
    const bool n_short = a.tiles_n <= a.tiles_k;
    const int ts = n_short ? a.tiles_n : a.tiles_k, tl = n_short ? a.tiles_k : a.tiles_n;
    const int full = tl / 4, bsz = ts * 4;
    int s_idx, l_idx;
    if (q < full * bsz) {
      const int inner = q % bsz;
      s_idx = inner / 4;
      l_idx = (q / bsz) * 4 + inner % 4;
    } else {
      const int rem = tl - full * 4, inner = q - full * bsz;
      s_idx = inner / rem;
      l_idx = full * 4 + inner % rem;
    }
    tn = n_short ? s_idx : l_idx;
    tk = n_short ? l_idx : s_idx;
  }
  if (split >= a.splits) return;
  const long row0 = (long)split * a.rows_per_split;
  const long row_end = min((long)a.M, row0 + a.rows_per_split);
  const int nstages = (int)((row_end - row0 + WG_BM - 1) / WG_BM);

  const int wm = wave >> 2, wn = wave & 3;
  f32x4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};

  const int li = lane & 15, q = li >> 2, p = li & 3, g = lane >> 4;
  const int row_off = (8 * g + q) * 512 + 8 * (p & 1);
  int ca[4], cb[4];   // swizzled 16-byte chunk offsets of the four 16-column sub-tiles
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ca[i] = (((8 * wm + 2 * i + (p >> 1)) ^ (q << 2) ^ ((g & 1) << 1)) << 4);
    cb[i] = (((8 * wn + 2 * i + (p >> 1)) ^ (q << 2) ^ ((g & 1) << 1)) << 4);
  }
  typedef short s4 __attribute__((ext_vector_type(4)));
  typedef short s8 __attribute__((ext_vector_type(8)));
  auto tr8 = [&](const char* base) {
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(base));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(base + 2048));   // rows + 4
    s8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return __builtin_bit_cast(bf16x8, f);
  };
  auto issue = [&](int st) {
    char* sa = smem + (st & 1) * 2 * WG_STAGE;
    const int lo = wg_opaque(lane);
    wg_fill<true>(sa, a.dy, a.ldy, tn * WG_TILE, a.N, row0 + (long)st * WG_BM, row_end, wave, lo);
    wg_fill<true>(sa + WG_STAGE, a.x, a.ldx, tk * WG_TILE, a.K, row0 + (long)st * WG_BM, row_end, wave, lo);
  };
  issue(0);
  for (int st = 0; st < nstages; ++st) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of stage st
    __syncthreads();
    if (st + 1 < nstages) issue(st + 1);
    char* sa = smem + wg_opaque((st & 1) * 2 * WG_STAGE) + row_off;
    char* sb = sa + WG_STAGE;
    const int valid = (int)min((long)WG_BM, row_end - (row0 + (long)st * WG_BM));
    const int ksteps = (valid + 31) / 32;
#pragma unroll 1
    for (int ks = 0; ks < ksteps; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tr8(sa + ks * 16384 + ca[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = tr8(sb + ks * 16384 + cb[j]);
      if ((ks + 1) * 32 > valid) {   // last stage of the last split: rows beyond the end repeat a valid row, zero them on the A side
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
          if (ks * 32 + 8 * g + jj >= valid) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i][jj] = (bf16_t)0.f;
          }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
  // acc[i][j][r] = C[n = 64 wm + 16 i + 4 (lane >> 4) + r][k = 64 wn + 16 j + (lane & 15)]
  const int n_pad = a.tiles_n * WG_TILE, k_pad = a.tiles_k * WG_TILE;
  float* wsb = a.ws + ((size_t)split * n_pad + (size_t)tn * WG_TILE) * k_pad + (size_t)tk * WG_TILE;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) wsb[(size_t)(64 * wm + 16 * i + 4 * g + r) * k_pad + 64 * wn + 16 * j + li] = acc[i][j][r];
}
#endif  // MMK_DEBUG_SWITCHES

// NARROW: instantiated for weights whose N or K is not a multiple of 256 -- only there can a wave's 64 x 64 block fall outside [N, K].
// The full-size shapes keep the loop without the test (with it they ran 1-4 % slower in a same-box A/B: the early `continue`
// changes the stage loop's code).
// TE: output tile edge, 256 (16 waves, one workgroup per CU) or 128 (4 waves, two per CU); per_xcd = workgroup slots of one XCD.
template <bool PAIR, bool NARROW = false, int TE = WG_TILE>
__global__ __launch_bounds__(64 * (TE / 64) * (TE / 64), TE == WG_TILE ? 1 : 2) void wgrad_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A | B]
  constexpr int WPR = TE / 64;              // waves per row of the wave grid
  constexpr int PITCH = TE * 2;             // bytes of a stage row
  constexpr int STAGE = WG_BM * PITCH;      // bytes of one operand stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware unit map: block b runs on XCD b % 8; XCD x owns splits x, x + 8, ...; consecutive slots of an XCD are the
  // tiles of one split
  const int T = a.tiles_n * a.tiles_k;
  int split, tn, tk;
  if (a.aligned) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    split = (slot / T) * 8 + xcd;
    const int tile = slot % T;
    tn = tile / a.tiles_k;
    tk = tile % a.tiles_k;
  } else {
    // more tiles than one XCD has CUs: consecutive positions of an XCD walk the tiles in blocks of (all tiles of the
    // short dimension) x (4 of the long one), so a block's 12-16 workgroups share each dY / x slab 3-4 ways in L2
    const int per_xcd = (a.splits * T + 7) >> 3;
    const int pos = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    split = pos / T;
    const int q = pos % T;
    const bool n_short = a.tiles_n <= a.tiles_k;
    const int ts = n_short ? a.tiles_n : a.tiles_k, tl = n_short ? a.tiles_k : a.tiles_n;  // short / long tile counts
    const int full = tl / 4, bsz = ts * 4;
    int s_idx, l_idx;
    if (q < full * bsz) {
      const int inner = q % bsz;
      s_idx = inner / 4;
      l_idx = (q / bsz) * 4 + inner % 4;
    } else {
      const int rem = tl - full * 4, inner = q - full * bsz;
      s_idx = inner / rem;
      l_idx = full * 4 + inner % rem;
    }
    tn = n_short ? s_idx : l_idx;
    tk = n_short ? l_idx : s_idx;
  }
  if (split >= a.splits) return;
  const long row0 = (long)split * a.rows_per_split;
  const long row_end = min((long)a.M, row0 + a.rows_per_split);
  const int nstages = (int)((row_end - row0 + WG_BM - 1) / WG_BM);

  // wave grid 4 x 4: output rows 64 wm.., output columns 64 wn...  A wave whose 64 x 64 block lies wholly beyond [N, K] (narrow
  // layers: a 96 x 96 weight fills 4 of the 16 blocks) takes part in the fills and barriers only -- it issues no fragment reads and
  // no MFMAs, and stores its zero accumulators into the workspace padding.  (Spreading a narrow tile's active waves over all four
  // SIMDs by another wave -> block map was tried: +1 % on the full-size shapes, nothing on the narrow ones, which are load-bound.)
  const int wm = wave / WPR, wn = wave % WPR;
  const bool active = !NARROW || (tn * TE + 64 * wm < a.N && tk * TE + 64 * wn < a.K);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // per-lane transposed-read offsets inside a stage (natural k order: element jj <-> stage row 16 ks + 8h + jj)
  const int li = lane & 15, q = li >> 2, p = li & 3, g1 = (lane >> 4) & 1, h = lane >> 5;
  // element (row, 16-B chunk c) of a stage sits at chunk position c ^ ((row & 3) << 2); for the 32-column tile t the
  // block's chunks are 4t + 2 g1 + (p >> 1), and row & 3 == q, so the swizzle turns into the tile index t ^ q
  int troff[2], xa[2], xb[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) troff[u] = (8 * h + 4 * u + q) * PITCH + ((2 * g1 + (p >> 1)) << 4) + 8 * (p & 1);
#pragma unroll
  for (int i = 0; i < 2; ++i) xa[i] = (((wm * 2 + i) ^ q) << 6);
#pragma unroll
  for (int j = 0; j < 2; ++j) xb[j] = (((wn * 2 + j) ^ q) << 6);
  typedef short s4 __attribute__((ext_vector_type(4)));
  typedef short s8 __attribute__((ext_vector_type(8)));
  auto tr8 = [&](const char* base) {
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(base));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(base + (troff[1] - troff[0])));
    s8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return __builtin_bit_cast(bf16x8, f);
  };

  auto issue = [&](int st) {
    char* sa = smem + (st & 1) * 2 * STAGE;
    const int lo = wg_opaque(lane);
    if constexpr (TE == WG_TILE) {
      wg_fill(sa, a.dy, a.ldy, tn * TE, a.N, row0 + (long)st * WG_BM, row_end, wave, lo);
      wg_fill(sa + STAGE, a.x, a.ldx, tk * TE, a.K, row0 + (long)st * WG_BM, row_end, wave, lo);
    } else {
      wg_fill_te<TE>(sa, a.dy, a.ldy, tn * TE, a.N, row0 + (long)st * WG_BM, row_end, wave, lo);
      wg_fill_te<TE>(sa + STAGE, a.x, a.ldx, tk * TE, a.K, row0 + (long)st * WG_BM, row_end, wave, lo);
    }
  };
  issue(0);
  for (int st = 0; st < nstages; ++st) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of stage st
    __syncthreads();                     // stage st complete; every wave is done with the other buffer
    if (st + 1 < nstages) issue(st + 1);
    if (NARROW && !active) continue;
    // per-stage base through an opaque value: the 12 per-lane fragment addresses are rebuilt here once per stage instead of
    // being hoisted for both buffers out of the loop (LDS offsets >= 64 KiB do not fit an instruction immediate, the
    // hoisted copies spilled)
    char* sa = smem + wg_opaque((st & 1) * 2 * STAGE) + troff[0];
    char* sb = sa + STAGE;
    const int valid = (int)min((long)WG_BM, row_end - (row0 + (long)st * WG_BM));
    // two 16-row steps per iteration, all sixteen fragment reads of the pair requested before its eight MFMAs: half as many
    // LDS waits per stage, and the second step's reads land behind the first step's MFMAs.  Every row of a stage is written by
    // the fill (rows past the split's end repeat a valid row), so a step beyond `valid` multiplies zeroed A rows by finite B rows.
    if (!PAIR) {   // A/B form (MMK_WGRAD_PAIR=0): one 16-row step per iteration, as in round 1
      const int ksteps = (valid + 15) / 16;
#pragma unroll 1
      for (int ks = 0; ks < ksteps; ++ks) {
        bf16x8 af[2], bfr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = tr8(sa + ks * (16 * PITCH) + xa[i]);
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[j] = tr8(sb + ks * (16 * PITCH) + xb[j]);
        if ((ks + 1) * 16 > valid) {
#pragma unroll
          for (int jj = 0; jj < 8; ++jj)
            if (ks * 16 + 8 * h + jj >= valid) {
#pragma unroll
              for (int i = 0; i < 2; ++i) af[i][jj] = (bf16_t)0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
      continue;
    }
    const int kpairs = (valid + 31) / 32;
#pragma unroll 1
    for (int kp = 0; kp < kpairs; ++kp) {
      bf16x8 af[2][2], bfr[2][2];
      const char* pa = sa + kp * (32 * PITCH);
      const char* pb = sb + kp * (32 * PITCH);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int i = 0; i < 2; ++i) af[u][i] = tr8(pa + u * (16 * PITCH) + xa[i]);
#pragma unroll
        for (int j = 0; j < 2; ++j) bfr[u][j] = tr8(pb + u * (16 * PITCH) + xb[j]);
      }
      if ((kp + 1) * 32 > valid) {  // last stage of the last split
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int jj = 0; jj < 8; ++jj)
            if ((2 * kp + u) * 16 + 8 * h + jj >= valid) {
#pragma unroll
              for (int i = 0; i < 2; ++i) af[u][i][jj] = (bf16_t)0.f;
            }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u][i], bfr[u][j], acc[i][j], 0, 0, 0);
    }
  }
  // ---- partial tile -> workspace: acc[i][j][e] = C[n = 64 wm + 32 i + (e&3) + 8(e>>2) + 4h][k = 64 wn + 32 j + (lane&31)]
  const int n_pad = a.tiles_n * TE, k_pad = a.tiles_k * TE;
  float* wsb = a.ws + ((size_t)split * n_pad + (size_t)tn * TE) * k_pad + (size_t)tk * TE;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = 64 * wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        wsb[(size_t)n * k_pad + 64 * wn + 32 * j + (lane & 31)] = acc[i][j][e];
      }
}

// ------------------------------------------------------------------------------------------------ the 8-phase form
// The schedule of the guide's "256^2 8-phase template" (cdna_hip_programming.md 5: 8 waves as 2 x 4, 128 x 64 of the output per wave,
// LDS-DMA prefetch that stays in flight across raw s_barriers behind a COUNTED vmcnt, the two waves of every SIMD a barrier apart so
// that one of them is always inside an MFMA cluster) for THIS product, whose operands both have the contraction along their rows:
// the fragments come from transposed reads (ds_read_b64_tr_b16) of [64 rows][128 columns] half-tile images instead of ds_read_b128.
//   * One 64-row step of the contraction ("K-tile") = four half-tiles of 16 KiB: A0 | A1 = columns 0..127 | 128..255 of the dY tile,
//     B0 | B1 likewise of the x tile; two K-tiles resident (128 KiB).  A wave (wm, wn) owns output rows 64 wm + {0..63} of BOTH A halves
//     and output columns 32 wn + {0..31} of BOTH B halves, i.e. four 64 x 32 quadrants (qm, qn), so that every half-tile is a contiguous
//     128-column slab of its operand (whole 128-byte lines per DMA lane group) and is last read in ONE known phase.
//   * Per K-tile four phases, each  { transposed reads of ONE half-tile | LDS-DMA of ONE half-tile, issued five phases before its
//     read | s_waitcnt vmcnt(8) -> s_barrier -> 8 x v_mfma_f32_32x32x16_bf16 -> s_barrier }:
//         phase 1  reads A0 (16)        MFMA (0,0)   stages B1 (t+1)        phase 3  reads A1 (16)        MFMA (1,1)   stages B0 (t+2)
//         phase 2  reads B1 (8)         MFMA (0,1)   stages A1 (t+1)        phase 4  reads B0 (t+1) (8)   MFMA (1,0)   stages A0 (t+2)
//     A fragments serve two quadrants from registers, B0 is read one phase early into a second register set (16 + 8 + 16 + 8 reads
//     instead of 24 + 8 + 16 + 0: no phase's reads outlast the partner wave's MFMA cluster): every LDS byte is read once per wave.
//   * vmcnt(8) after each phase's issue leaves the four youngest half-tiles (64 KiB per CU) in flight and retires exactly the one the
//     NEXT phase reads (read one phase after the wait that retires it); a slot is restaged three phases after its last read
//     (waves 4-7 run one barrier behind waves 0-3, so one phase is not enough).  Never vmcnt(0), never __syncthreads() in the loop.
// Serves weights whose N and K are multiples of 256; the others keep the two-stage kernel above.
constexpr int W8_HALF = 64 * 256;          // bytes of one half-tile image: 64 contraction rows x 128 columns bf16
constexpr int W8_BUF = 4 * W8_HALF;        // one K-tile: A0 | B0 | B1 | A1
constexpr int W8_A0 = 0, W8_B0 = W8_HALF, W8_B1 = 2 * W8_HALF, W8_A1 = 3 * W8_HALF;

__global__ __launch_bounds__(512, 1) void wgrad8_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 K-tiles][A0 | B0 | B1 | A1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.tiles_n * a.tiles_k;
  int split, tn, tk;
  if (a.aligned) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    split = (slot / T) * 8 + xcd;
    const int tile = slot % T;
    tn = tile / a.tiles_k;
    tk = tile % a.tiles_k;
  } else {   // (the unit map of wgrad_kernel: blocks of all tiles of the short dimension x 4 of the long one per XCD)
    const int per_xcd = (a.splits * T + 7) >> 3;
    const int pos = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    split = pos / T;
    const int q = pos % T;
    const bool n_short = a.tiles_n <= a.tiles_k;
    const int ts = n_short ? a.tiles_n : a.tiles_k, tl = n_short ? a.tiles_k : a.tiles_n;
    const int full = tl / 4, bsz = ts * 4;
    int s_idx, l_idx;
    if (q < full * bsz) {
      const int inner = q % bsz;
      s_idx = inner / 4;
      l_idx = (q / bsz) * 4 + inner % 4;
    } else {
      const int rem = tl - full * 4, inner = q - full * bsz;
      s_idx = inner / rem;
      l_idx = full * 4 + inner % rem;
    }
    tn = n_short ? s_idx : l_idx;
    tk = n_short ? l_idx : s_idx;
  }
  if (split >= a.splits) return;
  const long row0 = (long)split * a.rows_per_split;
  const int nrows = (int)(min((long)a.M, row0 + a.rows_per_split) - row0);   // contraction rows of this split
  const int nt = (nrows + WG_BM - 1) / WG_BM;

  const int wm = wave >> 2, wn = wave & 3;
  f32x16 acc[2][2][2];   // [qm][qn][mt]: rows 128 qm + 64 wm + 32 mt + .., columns 128 qn + 32 wn + ..
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][m][e] = 0.f;

  // transposed-read offsets inside a half-tile image (256-byte rows; chunk position = source chunk ^ ((row & 3) << 2), see wg_fill_te)
  const int li = lane & 15, q = li >> 2, p = li & 3, g1 = (lane >> 4) & 1, h = lane >> 5;
  const int tr0 = (8 * h + q) * 256 + ((2 * g1 + (p >> 1)) << 4) + 8 * (p & 1);
  const int offa0 = tr0 + (((2 * wm) ^ q) << 6), offa1 = tr0 + (((2 * wm + 1) ^ q) << 6), offb = tr0 + ((wn ^ q) << 6);
  typedef short s4 __attribute__((ext_vector_type(4)));
  typedef short s8 __attribute__((ext_vector_type(8)));
  auto tr8 = [&](const char* base) {   // rows +0..3 and +4..7 of one 16-row step
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(base));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(base + 1024));
    s8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return __builtin_bit_cast(bf16x8, f);
  };

  // LDS-DMA of one half-tile: 16 pieces of 1 KiB (4 rows x 256 B), two per wave; piece = rows 4 (2 wave + i) .. + 3, lane l -> row + (l >> 4),
  // chunk position l & 15 <- source chunk (l & 15) ^ (((l >> 4) & 3) << 2).  Rows past the split's end repeat its last row (finite
  // filler that the A side zeroes); whole K-tiles past the end are still issued (into a slot nobody reads again) so that the vmcnt
  // arithmetic is the same in every phase.
  const uint32_t smem_addr = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)smem);
  const int lrow = lane >> 4, lch = ((lane & 15) ^ ((lrow & 3) << 2)) * 8;
  // Full K-tiles: the per-lane byte offset (this wave's first piece inside a K-tile + row lrow of the piece + the swizzled source chunk)
  // never changes; a piece's address is ONE running scalar pointer per operand (K-tile t) plus small scalar constants -- a handful of
  // scalar adds per piece and no vector arithmetic beside the partner wave's MFMAs.
  const uint32_t voff_a = (uint32_t)((8 * wave + lrow) * (int)a.ldy + lch) * 2u, voff_b = (uint32_t)((8 * wave + lrow) * (int)a.ldx + lch) * 2u;
  const char* pa = wg_uniform(reinterpret_cast<const char*>(a.dy + row0 * a.ldy + tn * WG_TILE));   // K-tile t of this unit's dY / x columns
  const char* pb = wg_uniform(reinterpret_cast<const char*>(a.x + row0 * a.ldx + tk * WG_TILE));
  const int tile_a = 64 * (int)a.ldy * 2, tile_b = 64 * (int)a.ldx * 2;                   // bytes per K-tile (< 2^31: ld < 2^23)
  int rows_left = nrows;                                                                  // rows from K-tile t on
  // dt = K-tiles ahead of t (0 in the prologue, 1 or 2 in the loop)
  auto stage = [&](bool is_a, int colh, int dt, int slot_off) {
    const char* p0 = (is_a ? pa + (long)dt * tile_a : pb + (long)dt * tile_b) + colh * 2;
    const int ld = is_a ? (int)a.ldy : (int)a.ldx;
    if (rows_left >= 64 * dt + 64) {
      const uint32_t vo = is_a ? voff_a : voff_b;
      wg_dma16(p0, vo, smem_addr + slot_off + (2 * wave) * 1024);
      wg_dma16(p0 + 8 * ld, vo, smem_addr + slot_off + (2 * wave + 1) * 1024);
    } else {   // the split's last, ragged K-tile and the run-ahead tiles past it: clamp every row to the split's last one
      const int last = rows_left - 1 - 64 * dt;                 // last valid row relative to the K-tile's first (may be negative)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int pr = 4 * (2 * wave + i);                      // first row of the piece
        const int srow = min(pr, last);                         // wave-uniform, may be negative (a K-tile wholly past the end): SCALAR part
        const int drow = min(pr + lrow, last) - srow;           // 0 .. 3: the per-lane offset stays non-negative (it is zero-extended)
        wg_dma16(p0 + (long)srow * ld * 2, (uint32_t)(drow * ld + lch) * 2u, smem_addr + slot_off + (2 * wave + i) * 1024);
      }
    }
  };
  bf16x8 af[2][4], b0[2][4], b1[4];   // b0[K-tile parity]: the next K-tile's B0 arrives while this one's is still in use
  auto zero_tail = [&](int valid) {   // rows >= valid of the last K-tile: zero on the A side (B rows are finite filler)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
        if (16 * ks + 8 * h + jj >= valid) {
          af[0][ks][jj] = (bf16_t)0.f;
          af[1][ks][jj] = (bf16_t)0.f;
        }
  };
  auto mfma8 = [&](f32x16 (&c)[2], const bf16x8 (&bb)[4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      c[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][ks], bb[ks], c[0], 0, 0, 0);
      c[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][ks], bb[ks], c[1], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // diagnostic build: shader-clock sums of the four segments of a phase (MFMA cluster, wait at the closing barrier, load segment,
  // wait at the opening barrier), per wave of workgroup 0; read SHARES from it, never the run time (guide 7, In-kernel stamps)
#ifdef MMK_WGRAD_STAMPS_BUILD
  unsigned long long st_sum[4] = {0, 0, 0, 0}, st_last = 0, st_first = 0, rt_first = 0;
#define W8_STAMP(i)                                                                              \
  {                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    unsigned long long _t;                                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                   \
    if ((i) >= 0) st_sum[(i) < 0 ? 0 : (i)] += _t - st_last;                                     \
    st_last = _t;                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  }
#else
#define W8_STAMP(i)
#endif
#define W8_WAIT_BAR()                                   \
  __builtin_amdgcn_s_waitcnt(0x0F78); /* vmcnt(8) */    \
  W8_STAMP(2)                                           \
  __builtin_amdgcn_s_barrier();                         \
  W8_STAMP(3)                                           \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
  __builtin_amdgcn_sched_barrier(0)
#define W8_END_BAR()            \
  W8_STAMP(0)                   \
  __builtin_amdgcn_s_barrier(); \
  W8_STAMP(1)
  // one K-tile in buffer BUF (compile-time LDS offsets): t = its index
  auto ktile = [&](auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    const char* img = smem + BUF * W8_BUF;
    const char* nxt = smem + (BUF ^ 1) * W8_BUF;
    const int valid = rows_left;
    // ---- phase 1: A0 -> quadrant (0, 0); stage B1 (t + 1)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      af[0][ks] = tr8(img + W8_A0 + ks * 4096 + offa0);
      af[1][ks] = tr8(img + W8_A0 + ks * 4096 + offa1);
    }
    stage(false, 128, 1, (BUF ^ 1) * W8_BUF + W8_B1);
    W8_WAIT_BAR();
    if (valid < 64) zero_tail(valid);
    mfma8(acc[0][0], b0[BUF]);
    W8_END_BAR();
    // ---- phase 2: B1 -> quadrant (0, 1); stage A1 (t + 1)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b1[ks] = tr8(img + W8_B1 + ks * 4096 + offb);
    stage(true, 128, 1, (BUF ^ 1) * W8_BUF + W8_A1);
    W8_WAIT_BAR();
    mfma8(acc[0][1], b1);
    W8_END_BAR();
    // ---- phase 3: A1 -> quadrant (1, 1); stage B0 (t + 2)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      af[0][ks] = tr8(img + W8_A1 + ks * 4096 + offa0);
      af[1][ks] = tr8(img + W8_A1 + ks * 4096 + offa1);
    }
    stage(false, 0, 2, BUF * W8_BUF + W8_B0);
    W8_WAIT_BAR();
    if (valid < 64) zero_tail(valid);
    mfma8(acc[1][1], b1);
    W8_END_BAR();
    // ---- phase 4: B0 of the NEXT K-tile into the other register set; quadrant (1, 0) from registers; stage A0 (t + 2)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b0[BUF ^ 1][ks] = tr8(nxt + W8_B0 + ks * 4096 + offb);
    stage(true, 0, 2, BUF * W8_BUF + W8_A0);
    W8_WAIT_BAR();
    mfma8(acc[1][0], b0[BUF]);
    W8_END_BAR();
    pa = wg_uniform(pa + tile_a);
    pb = wg_uniform(pb + tile_b);
    rows_left -= 64;
  };
  // ---- prologue: the six half-tiles whose reads come first, in the order the phases will keep issuing
  stage(false, 0, 0, W8_B0);
  stage(true, 0, 0, W8_A0);
  stage(false, 128, 0, W8_B1);
  stage(true, 128, 0, W8_A1);
  stage(false, 0, 1, W8_BUF + W8_B0);
  stage(true, 0, 1, W8_BUF + W8_A0);
  __builtin_amdgcn_s_waitcnt(0x0F78);   // B0, A0 of K-tile 0 (this wave's pieces)
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) b0[0][ks] = tr8(smem + W8_B0 + ks * 4096 + offb);
  if (wave >= 4) __builtin_amdgcn_s_barrier();   // the second wave of every SIMD runs one barrier behind the first
#ifdef MMK_WGRAD_STAMPS_BUILD
  W8_STAMP(-1)
  st_first = st_last;
  rt_first = __builtin_amdgcn_s_memrealtime();
#endif
  int t = 0;
#pragma unroll 1
  for (; t + 1 < nt; t += 2) {
    ktile(std::integral_constant<int, 0>{});
    ktile(std::integral_constant<int, 1>{});
  }
  if (t < nt) ktile(std::integral_constant<int, 0>{});
#ifdef MMK_WGRAD_STAMPS_BUILD
  if (a.stamps != nullptr && blockIdx.x == 0 && lane == 0) {
    unsigned long long* o = a.stamps + wave * 8;
    o[0] = st_sum[0]; o[1] = st_sum[1]; o[2] = st_sum[2]; o[3] = st_sum[3];
    o[4] = st_last - st_first;
    o[5] = __builtin_amdgcn_s_memrealtime() - rt_first;   // 100 MHz ticks
    o[6] = (unsigned long long)nt;
  }
#endif
  if (wave < 4) __builtin_amdgcn_s_barrier();    // pairs with the extra barrier of waves 4-7
  __builtin_amdgcn_s_waitcnt(0x0F70);            // drain the (unused) run-ahead DMAs before the LDS is released
#undef W8_WAIT_BAR
#undef W8_END_BAR
#undef W8_STAMP
  // ---- partial tile -> workspace: acc[qm][qn][mt][e] = C[n = 128 qm + 64 wm + 32 mt + (e&3) + 8(e>>2) + 4h][k = 128 qn + 32 wn + (lane&31)]
  const int n_pad = a.tiles_n * WG_TILE, k_pad = a.tiles_k * WG_TILE;
  float* wsb = a.ws + ((size_t)split * n_pad + (size_t)tn * WG_TILE) * k_pad + (size_t)tk * WG_TILE;
#pragma unroll
  for (int qm = 0; qm < 2; ++qm)
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int n = 128 * qm + 64 * wm + 32 * mt + (e & 3) + 8 * (e >> 2) + 4 * h;
          wsb[(size_t)n * k_pad + 128 * qn + 32 * wn + (lane & 31)] = acc[qm][qn][mt][e];
        }
}

}  // namespace mmk

using namespace mmk;

extern "C" {

// plan: number of M splits and workspace floats for a [N, K] weight gradient over M rows
int mmk_wgrad_plan(int64_t M, int N, int K, int* splits_out, int64_t* ws_floats_out) {
  MMK_REQUIRE(M > 0 && N > 0 && K > 0 && splits_out && ws_floats_out, "bad arguments");
  const int te = wg_tile_edge(N, K);
  const int tn = cdiv(N, te), tk = cdiv(K, te), T = tn * tk;
  // one workgroup per CU and a single round: with T <= 32 tiles every XCD (32 CUs) hosts floor(32 / T) whole splits.
  // MMK_WGRAD_RESERVE_CUS (per XCD, default 0) leaves CUs to a concurrent kernel such as RCCL's all-reduce -- a launch
  // that no longer fits one round takes two -- at +8 % for the 36-tile shapes when the GPU is not shared.
  static const int reserve = getenv("MMK_WGRAD_RESERVE_CUS") ? std::min(16, std::max(0, atoi(getenv("MMK_WGRAD_RESERVE_CUS")))) : 0;
  const int per_xcd = (te == WG_TILE ? 32 : 64) - (te == WG_TILE ? reserve : 2 * reserve);   // 128-tiles: two workgroups per CU
  int splits, aligned;
  wg_plan_units(T, per_xcd, &splits, &aligned);
  splits = (int)std::min<int64_t>(splits, std::max<int64_t>(1, M / 512));
  *splits_out = splits;
  *ws_floats_out = (int64_t)splits * tn * te * tk * te;
  return 0;
}

static int wgrad_launch(const void* dy, const void* x, float* ws, int64_t M, int N, int K, int64_t ldy, int64_t ldx, WgradArgs* out,
                        hipStream_t st);

// diagnostic builds (-DMMK_WGRAD_STAMPS_BUILD, MMK_WGRAD_STAMPS=1): 8 waves x 8 words that workgroup 0 of the 8-phase kernel fills
static unsigned long long* wgrad_stamp_buffer() {
#ifdef MMK_WGRAD_STAMPS_BUILD
  static unsigned long long* buf = nullptr;
  static bool tried = false;
  if (!tried) {
    tried = true;
    if (getenv("MMK_WGRAD_STAMPS") != nullptr && hipMalloc(reinterpret_cast<void**>(&buf), 64 * sizeof(unsigned long long)) != hipSuccess) buf = nullptr;
  }
  return buf;
#else
  return nullptr;
#endif
}
int mmk_wgrad_debug_stamps(unsigned long long* out) {
  unsigned long long* buf = wgrad_stamp_buffer();
  MMK_REQUIRE(buf != nullptr && out != nullptr, "no stamp buffer (needs a library built with -DMMK_WGRAD_STAMPS_BUILD and MMK_WGRAD_STAMPS=1)");
  MMK_HIP(hipDeviceSynchronize());
  MMK_HIP(hipMemcpy(out, buf, sizeof(unsigned long long) * 64, hipMemcpyDeviceToHost));
  return 0;
}

// The split partial tiles only (no reduction): ws[split][n_pad][k_pad] f32 with n_pad / k_pad = N / K rounded up to 256;
// the caller sums the splits.  Used by the contrastive loss' backward (csrc/clip.hip): dB = G^T A is this kernel's
// "both operands contracted along their rows" form, so G^T never has to exist.
int mmk_wgrad_partial(const void* dy, const void* x, float* ws, int64_t M, int N, int K, int64_t ldy, int64_t ldx, int32_t* splits_out,
                      int32_t* n_pad_out, int32_t* k_pad_out, void* stream) {
  MMK_REQUIRE(dy && x && ws && M > 0 && N > 0 && K > 0, "bad arguments");
  MMK_REQUIRE(N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0, "wgrad: N, K and the row strides must be multiples of 8");
  WgradArgs a;
  int rc = wgrad_launch(dy, x, ws, M, N, K, ldy, ldx, &a, static_cast<hipStream_t>(stream));
  if (rc) return rc;
  if (splits_out) *splits_out = a.splits;
  const int te = wg_tile_edge(N, K);
  if (n_pad_out) *n_pad_out = a.tiles_n * te;
  if (k_pad_out) *k_pad_out = a.tiles_k * te;
  return 0;
}

static int wgrad_launch(const void* dy, const void* x, float* ws, int64_t M, int N, int K, int64_t ldy, int64_t ldx, WgradArgs* out,
                        hipStream_t st) {
  WgradArgs a;
  a.dy = static_cast<const bf16_t*>(dy); a.x = static_cast<const bf16_t*>(x); a.ws = ws;
  a.ldy = ldy; a.ldx = ldx; a.M = (int)M; a.N = N; a.K = K;
  a.stamps = wgrad_stamp_buffer();
  const int te = wg_tile_edge(N, K);
  a.tiles_n = cdiv(N, te); a.tiles_k = cdiv(K, te);
  int64_t wsf;
  mmk_wgrad_plan(M, N, K, &a.splits, &wsf);
  a.rows_per_split = round_up((int)cdiv((int)M, a.splits), WG_BM);
  a.splits = cdiv((int)M, a.rows_per_split);
  const int T = a.tiles_n * a.tiles_k;
  {
    int s_unused;
    wg_plan_units(T, te == WG_TILE ? 32 : 64, &s_unused, &a.aligned);
  }
  const int grid = a.aligned ? 8 * cdiv(a.splits, 8) * T : 8 * cdiv(a.splits * T, 8);
  const int bytes = 4 * WG_BM * te * 2;
  const int threads = 64 * (te / 64) * (te / 64);
  static const bool pair = !(MMK_DBG_ENV("MMK_WGRAD_PAIR") && atoi(MMK_DBG_ENV("MMK_WGRAD_PAIR")) == 0);
#ifdef MMK_DEBUG_SWITCHES
  // MFMA shape experiment: MMK_WGRAD_MFMA=16 selects the 16x16x32 kernel (read per call: interleaved A/B runs)
  const bool mfma16 = MMK_DBG_ENV("MMK_WGRAD_MFMA") && atoi(MMK_DBG_ENV("MMK_WGRAD_MFMA")) == 16;
#endif
  // the variant this shape runs; its > 64 KiB LDS opt-in is made per device (kernel_setup)
  const bool ragged = N % te != 0 || K % te != 0;
  const void* kern = pair && ragged ? reinterpret_cast<const void*>(wgrad_kernel<true, true>)
                     : pair         ? reinterpret_cast<const void*>(wgrad_kernel<true>)
                                    : reinterpret_cast<const void*>(wgrad_kernel<false>);
  if (te == WG_TILE_NARROW)
    kern = ragged ? reinterpret_cast<const void*>(wgrad_kernel<true, true, WG_TILE_NARROW>)
                  : reinterpret_cast<const void*>(wgrad_kernel<true, false, WG_TILE_NARROW>);
#ifdef MMK_DEBUG_SWITCHES
  if (mfma16 && te == WG_TILE) kern = reinterpret_cast<const void*>(wgrad_kernel16);
#endif
  int threads_l = threads;
  // the 8-phase kernel: full 256-tiles only (N, K multiples of 256) and splits long enough for its two-K-tile prologue to pay
  bool eight = kWgrad8Default;
  if (const char* e = MMK_DBG_ENV("MMK_WGRAD_KERNEL")) eight = atoi(e) == 8;   // A/B in debug-switch builds, read per call
  if (eight && te == WG_TILE && !ragged && a.rows_per_split >= 4 * WG_BM) {
    kern = reinterpret_cast<const void*>(wgrad8_kernel);
    threads_l = 512;
  }
  KernelSetup ks;
  if (int rc = kernel_setup(kern, threads_l, bytes, &ks)) return rc;
  {
    ProfEvents pe(MMK_K_WGRAD);
    void* params[] = {&a};
    MMK_HIP(hipExtLaunchKernel(kern, dim3(grid), dim3(threads_l), params, bytes, st, pe.start, pe.stop, 0));
  }
  MMK_LAUNCH_CHECK();
  *out = a;
  return 0;
}

int mmk_wgrad(const void* dy, const void* x, void* dw, float* ws, int64_t M, int N, int K, int64_t ldy, int64_t ldx, int64_t ldw,
              int out_dtype, void* stream) {
  MMK_REQUIRE(dy && x && dw && ws && M > 0 && N > 0 && K > 0, "bad arguments");
  MMK_REQUIRE(N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0 && ldw % 4 == 0, "wgrad: N, K and the row strides must be multiples of 8");
  WgradArgs a;
  hipStream_t st = static_cast<hipStream_t>(stream);
  {
    int rc = wgrad_launch(dy, x, ws, M, N, K, ldy, ldx, &a, st);
    if (rc) return rc;
  }
  const long n4 = (long)N * (K / 4);
  const int te_ = wg_tile_edge(N, K);
  const int n_pad = a.tiles_n * te_, k_pad = a.tiles_k * te_;
  int rc = MMK_DISPATCH_DTYPE(out_dtype, OUT, [&]() -> int {
    if (a.splits >= 32)
      hipLaunchKernelGGL((wgrad_reduce_wide_kernel<OUT>), dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, ws, static_cast<OUT*>(dw),
                         (long)ldw, N, K, n_pad, k_pad, a.splits);
    else
      hipLaunchKernelGGL((wgrad_reduce_kernel<OUT>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, ws, static_cast<OUT*>(dw),
                         (long)ldw, N, K, n_pad, k_pad, a.splits);
    return 0;
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}
}
