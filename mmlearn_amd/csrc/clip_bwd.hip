// (This file also holds the forward statistics of the same directions as one streaming launch: clip_fwd_shard_kernel, below.)
// Row-sharded CLIP / InfoNCE backward as ONE kernel that recomputes G: for a direction whose rows are a SHARD of the batch (R owned
// rows against C >> R gathered columns: what every rank runs at W > 1) the two launches
//     sim_grad   S = X Y^T again, G = c_row P_row + c_col P_col - c_diag delta -> HBM (R x C bf16)
//     grad_gemm  dX = G Y (split over C into f32 slabs), reading G back and a TRANSPOSED copy of Y
// become one.  A workgroup (4 waves, one per SIMD) owns 64 owned rows x one column split and walks the split in tiles of 64 columns:
//     phase S   wave (wit, wjt) computes S^T[32 j x 32 i] over the WHOLE contraction (D = 512: 32 MFMAs); its X fragments stay in
//               REGISTERS for the whole kernel (128 VGPRs), the Y tile is an LDS image filled by LDS-DMA, read as row fragments that
//               run eight k steps ahead of their MFMAs; the NEXT tile's DMA pieces are issued between the MFMAs
//     G         each lane finishes 16 logits of one owned row: P_row, P_col from the row / column log-sum-exps (log2 domain), G, the
//               d/dscale term; G rounded to bf16 into an 8-KiB [64 i][64 j] image.  Packed f32 arithmetic, no branches
//     phase D   dX[64 i x 512] += G Y_tile: G row fragments x TRANSPOSED reads (ds_read_b64_tr_b16) of the SAME Y image -- no
//               transposed copy of Y exists, G never leaves the chip; a wave owns 128 of the 512 output columns (8 accumulator tiles)
// The f32 tile goes to the split's slab; grad_finalize (clip.hip) sums the slabs as before.  One image format serves the row reads of
// phase S and the transposed reads of phase D: the [32 rows][64 k] sub-images of csrc/attention.hip (128-byte rows, chunk ^ swz(row)).
// Two barriers per 64 columns.  LDS: two 64-KiB Y tiles + the G image = 137 KiB, so ONE workgroup per CU and one wave per SIMD:
// nothing overlaps a wave's own latencies, which is why the reads, waits and DMA issue are placed by hand.  No single unit bounds
// it (rocprofv3 --pmc, R = 1024 x C = 8192: MFMA pipe busy 33 % of the kernel, LDS array 20 % -- 82 LDS instructions per wave and
// tile, 15 % of their cycles bank conflicts --, ~4,500 VALU instructions per wave ~25 %, waits 29 % of the wave cycles): what is
// left is the serial order of a tile's phases on each SIMD.  A variant that ran the G arithmetic of tile t between the dX MFMAs of
// tile t - 1 (their operands read into registers at the top of the tile, one barrier per tile, MFMAs as inline instructions with
// fixed register files) measured the same 47 us and was dropped: it trades the exposed arithmetic for 40 exposed LDS reads at the
// top of every tile and a four-deep read ring in phase S (registers).  profiles/r06_loss_shard_fused_bwd.json keeps that and the
// first form of this kernel (32-column tiles with the contraction split over the waves and an LDS reduction of the partial tiles:
// 101 us, slower than the two launches' 69 us).
// HBM sees the operands once per (split, row-block group on one XCD) plus the slabs: no G (2 x 33 MB at R = 1024 x C = 8192), no
// transposed Y.  Eligibility (clip.hip): bf16 compute, k_pad = 512, a direction with a tile pass of its own whose G nobody else reads.
#include <hip/hip_ext.h>

#include <algorithm>
#include <type_traits>

#include "clip_internal.h"
#include "common.h"

namespace mmk {

constexpr int CB_ROWS = 64;        // owned rows per workgroup
constexpr int CB_JT = 64;          // columns per tile
constexpr int CB_KP = 512;         // k_pad served
constexpr int CB_SUB = 32 * 128;               // bytes of one [32 rows][64 k] sub-image
constexpr int CB_YBUF = 2 * (CB_KP / 64) * CB_SUB;   // one Y tile: 2 (column halves) x 8 (k) sub-images = 64 KiB
constexpr int CB_GIMG = CB_ROWS * 128;             // G image: [64 i][64 j] bf16 = 8 KiB
constexpr int CB_LCOL = 256;                       // one tile's column log-sum-exps: 64 floats = one 64-lane x 4-byte LDS-DMA piece
constexpr int CB_LDS = 2 * CB_YBUF + CB_GIMG + 2 * CB_LCOL;

// (the image helpers of csrc/attention.hip, restated for this translation unit)
__device__ __forceinline__ int cb_swz(int row) {
  const int t = row >> 1;
  return ((t & 1) << 2) | (t & 2) | ((t >> 2) & 1);
}
__device__ __forceinline__ void cb_dma16(const void* sbase, uint32_t voff, uint32_t lds_addr) {
  const uint64_t pb = reinterpret_cast<uint64_t>(sbase);
  const uint64_t ps = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pb >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pb);
  lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(reinterpret_cast<const void*>(ps)), "s"(lds_addr) : "memory");
}

__device__ __forceinline__ void cb_dma4(const void* sbase, uint32_t voff, uint32_t lds_addr) {   // 64 lanes x 4 bytes
  const uint64_t pb = reinterpret_cast<uint64_t>(sbase);
  const uint64_t ps = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pb >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pb);
  lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(reinterpret_cast<const void*>(ps)), "s"(lds_addr) : "memory");
}

__global__ __launch_bounds__(256) void clip_bwd_fused_kernel(const BwdFusedBatch batch, const float* __restrict__ scale_ptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ybuf = smem;                                               // [2][2 jt][8 k sub-images][32 rows][128 B]
  char* gimg = smem + 2 * CB_YBUF;                                 // [64 i][128 B], 16-byte chunk ^= i & 7
  const float* lcol = reinterpret_cast<const float*>(gimg + CB_GIMG);   // [2][64]: lse_col of the tile's 64 columns
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wit = wave & 1, wjt = wave >> 1;   // phase S: this wave's 32 x 32 tile of the 64 x 64 logits
  // unit map: workgroups are dealt to the XCDs round-robin; all row blocks of a (problem, split) unit read the same 1 MiB of Y, so a
  // unit lives on ONE XCD (speed only)
  // (with fewer than 8 units the row blocks of a unit are dealt to `groups` XCDs, so that all eight are busy)
  const int lin = blockIdx.x, xcd = lin & 7, jj_ = lin >> 3;
  const int vunit = xcd + 8 * (jj_ / batch.rows_per_group);
  const int unit = vunit / batch.groups, rb = (vunit % batch.groups) * batch.rows_per_group + jj_ % batch.rows_per_group;
  if (unit >= batch.n_probs * batch.n_split || rb >= batch.row_blocks) return;
  const int split = unit % batch.n_split;
  const BwdFusedProb& p = batch.p[unit / batch.n_split];
  const int i0 = rb * CB_ROWS;
  if (i0 >= p.r) return;
  // (the problem's scalars as values: fields read through the reference become scalar loads wherever they are used -- inside the loop)
  const int n_cols = p.c;
  const bf16_t* y_rows = p.y;
  const float* lse_col = p.lse_col;
  asm volatile("" : "+s"(y_rows), "+s"(lse_col));   // (opaque: otherwise re-loaded from the argument block inside the loop, a scalar load
                                                   // whose wait -- lgkmcnt(0) -- lands between the counted LDS waits of phase S)
  const float c_row = p.c_row, c_col = p.c_col, c_diag = p.c_diag, s_row = p.s_row, s_col = p.s_col, s_diag = p.s_diag;   // rows beyond r are never read by the finalize
  const int c0 = split * batch.cols_per_split;
  // tiles that begin inside the c columns (a tile of padding only contributes nothing); the last one may be ragged
  const int ntile = max(0, (min(n_cols, c0 + batch.cols_per_split) - c0 + CB_JT - 1) / CB_JT);
  const float s2 = *scale_ptr * 1.4426950408889634f;
  const bool use_col = (c_col != 0.f) || (s_col != 0.f);
  const int dbg = kDebugSwitches ? batch.dbg : 0;   // timing ablations (debug-switch builds, MMK_CB_DBG; WRONG results): 1 no phase S, 2 no G
                                                    // arithmetic, 4 no phase D, 8 no Y DMA, 16 no slab stores

  // ---- per-lane LDS offsets.  Row fragment kk of image row r (k = 16 kk + 8 h ..): r * 128 + (((2 kk + h) ^ swz(r)) << 4).
  // Transposed fragment in NATURAL k order (element jj <-> image row 16 js + 8 h + jj) of column tile ct: two ds_read_b64_tr_b16,
  // lane 4 q + pp of a 16-lane group addresses row 8 h + 4 u + q, columns 4 pp .. + 3 of the group's 16 (chunk 4 ct + 2 g1 + (pp >> 1)).
  int rowoff[4], troff[2][2];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) rowoff[kk] = r * 128 + (((2 * kk + h) ^ cb_swz(r)) << 4);
  {
    const int li = lane & 15, q = li >> 2, pp = li & 3, g1 = (lane >> 4) & 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int row = 8 * h + 4 * u + q;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) troff[u][ct] = row * 128 + (((4 * ct + 2 * g1 + (pp >> 1)) ^ cb_swz(row)) << 4) + 8 * (pp & 1);
    }
  }
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

  // ---- Y tile t -> buffer b: 64 pieces of 1 KiB (8 rows x 128 B of one sub-image), sixteen per wave (k sub-images 2 wave, 2 wave + 1 of
  // both column halves); the swizzle goes on the SOURCE chunk.  Only rows the descriptor promises are read (x: r rows, y: c rows --
  // include/mmlearn_hip.h; a shard is often a SLICE of the gathered operand that ends with it): a row past the last one reads the
  // last one again (`avail` = last valid row - first row of the tile, >= 0), and the G arithmetic zeroes what comes of it.
  const uint32_t smem_addr = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)smem);
  auto issue_piece = [&](const bf16_t* base, int avail, int b, int idx) {   // piece idx (0..15) of this wave's share of 64 rows x 512 from base -> buffer b
    const int jt = idx >> 3, sub = (idx >> 2) & 1, q8 = idx & 3;
    const int s = 2 * wave + sub;
    const int row = 8 * q8 + (lane >> 3);
    const int ch = (lane & 7) ^ cb_swz(row);
    cb_dma16(base + 64 * s, (uint32_t)(min(32 * jt + row, avail) * CB_KP + ch * 8) * 2u,
             smem_addr + b * CB_YBUF + (8 * jt + s) * CB_SUB + q8 * 1024);
  };
  auto issue_rows = [&](const bf16_t* base, int avail, int b) {   // 64 rows x 512 from base -> buffer b
#pragma unroll
    for (int idx = 0; idx < 16; ++idx) issue_piece(base, avail, b, idx);
  };
  // the tile's column log-sum-exps ride along as one more piece (wave 0): a plain load inside the loop would make the compiler wait
  // for ALL outstanding vector memory operations -- the next tile's pieces included -- at its first use
  auto issue_lcol = [&](int t, int b) {
    if (use_col && wave == 0)
      cb_dma4(lse_col, (uint32_t)min(c0 + CB_JT * t + lane, n_cols - 1) * 4u, smem_addr + 2 * CB_YBUF + CB_GIMG + b * CB_LCOL);
  };
  auto issue_y = [&](int t, int b) {
    issue_rows(y_rows + (long)(c0 + CB_JT * t) * CB_KP, n_cols - 1 - (c0 + CB_JT * t), b);
    issue_lcol(t, b);
  };

  f32x16 dacc[2][4];   // [it][kt]: rows i0 + 32 it + .., columns 128 wave + 32 kt + ..
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) dacc[it][kt][e] = 0.f;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 ds2 = {0.f, 0.f};
  // the G step's constants: a lane finishes the logits of ONE owned row (32 wit + r) against 16 columns of the wave's 32
  const int ri = 32 * wit + r, iglob = i0 + ri;
  const bool ivalid = iglob < p.r;
  const float lr2 = (ivalid ? p.lse_row[iglob] : 0.f) * 1.4426950408889634f;
  const int lab = p.label_off + iglob;
  const int lab_lo = p.label_off + i0;          // this block's label columns: lab_lo .. lab_lo + 63
  const bool rows_full = i0 + CB_ROWS <= p.r;
  if (!use_col && tid < 2 * (CB_LCOL / 4)) const_cast<float*>(lcol)[tid] = 1e30f;   // exp2(-1e30 log2 e) = 0: P_col vanishes without a select

  // ---- this wave's X fragments: B operand of phase S, rows i0 + 32 wit + r, k = 16 ks + 8 h .. + 7, the whole contraction, in
  // registers for the whole kernel.  The block is a 64 x 512 tile like any Y tile: it comes through the second buffer as an image
  // (coalesced 128-byte rows by LDS-DMA; row-strided 32-byte loads straight into registers cost several microseconds here), while
  // the first Y tile is already on its way into the first.
  if (ntile > 0 && !(dbg & 8)) issue_y(0, 0);
  issue_rows(p.x + (long)i0 * CB_KP, p.r - 1 - i0, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bf16x8 xf[32];
#pragma unroll
  for (int ks = 0; ks < 32; ++ks)
    xf[ks] = *reinterpret_cast<const bf16x8*>(ybuf + CB_YBUF + (8 * wit + (ks >> 2)) * CB_SUB + rowoff[ks & 3]);
  // (consumed here: the reads are over before this wave reaches the loop's first barrier, after which the buffer is overwritten)
#pragma unroll
  for (int ks = 0; ks < 32; ++ks) asm volatile("" : "+v"(xf[ks]));
#pragma unroll 1
  for (int t = 0; t < ntile; ++t) {
    const char* yb = ybuf + (t & 1) * CB_YBUF;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t
    __syncthreads();                                   // tile t complete; phase D of tile t - 1 is over: the other buffer and the G image are free
    // tile t + 1 goes out piece by piece BETWEEN the MFMAs of phase S (issued in one burst ahead of them the ~100 scalar and DMA
    // instructions cost this wave ~0.2 us per tile with nothing else to run on its SIMD).  Past the last tile the pieces re-load the
    // last tile into the free buffer (unused) rather than branch inside the MFMA sequence.
    const int tn = min(t + 1, ntile - 1), nb = (t + 1) & 1;
    const bf16_t* ynext = y_rows + (long)(c0 + CB_JT * tn) * CB_KP;
    const int avail_next = n_cols - 1 - (c0 + CB_JT * tn);
    // ---------------- phase S: S^T[32 j (wjt)][32 i (wit)] over the whole contraction, two accumulators (even / odd k steps)
    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
    if (!(dbg & 1)) {
      // The Y row fragments run eight k steps ahead of their MFMAs (one wave per SIMD: nobody else hides the LDS latency), with the
      // next tile's DMA pieces between them.  Reads and waits are spelled out: left to the scheduler the reads sink to just before
      // their MFMA.  LDS operations return in order, so before MFMA ks the reads younger than fragment ks may be outstanding:
      // min(ks + 8, 32) - (ks + 1) of them.
      uint32_t ya[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) ya[kk] = smem_addr + (t & 1) * CB_YBUF + 8 * wjt * CB_SUB + rowoff[kk];
      bf16x8 yf[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(yf[ks]) : "v"(ya[ks & 3]), "n"((ks >> 2) * CB_SUB) : "memory");
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) {
        if (ks <= 24) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(yf[ks & 7]) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(yf[ks & 7]) : "n"(31 - ks) : "memory");
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[ks & 7], xf[ks], sacc, 0, 0, 0);
        // (the fragment register is re-loaded right behind the MFMA that reads it: the MFMA has taken its operands by then -- issue is
        // in order and the hardware interlocks a VGPR that a pending MFMA still has to read)
        if (ks + 8 < 32)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(yf[ks & 7]) : "v"(ya[ks & 3]), "n"(((ks + 8) >> 2) * CB_SUB) : "memory");
        if (ks & 1) issue_piece(ynext, avail_next, nb, ks >> 1);
      }
      issue_lcol(tn, nb);
    }
    // ---------------- G: element e of the tile is logit (row ri, column 32 wjt + (e&3) + 8 (e>>2) + 4 h).  Branch-free (selects
    // only), and two forms: a tile that holds no label column, no column beyond c and no row beyond r -- all but a few -- skips the
    // diagonal and validity selects (about 10 VALU operations + 2 exp per element; nothing overlaps them at one wave per SIMD).
    if (!(dbg & 2)) {
      const float* lc = lcol + (t & 1) * (CB_LCOL / 4) + 32 * wjt + 4 * h;
      const int jt0 = c0 + CB_JT * t;
      const f32x2 s2v = {s2, s2}, nlr2v = {-lr2, -lr2}, nlog2e = {-1.4426950408889634f, -1.4426950408889634f};
      const f32x2 c_rowv = {c_row, c_row}, c_colv = {c_col, c_col}, s_rowv = {s_row, s_row}, s_colv = {s_col, s_col};
      auto g_step = [&](auto general) {
        constexpr bool GEN = decltype(general)::value;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(lc + 8 * q4);
          bf16x4 g4;
#pragma unroll
          for (int u2 = 0; u2 < 2; ++u2) {   // two elements at a time: packed f32 multiply-adds (v_pk_fma_f32), half the VALU issue slots
            const int e = 4 * q4 + 2 * u2;
            const f32x2 tv = {sacc[e], sacc[e + 1]};
            const f32x2 nl = {l4[2 * u2], l4[2 * u2 + 1]};
            const f32x2 ar = __builtin_elementwise_fma(tv, s2v, nlr2v);
            const f32x2 ac = __builtin_elementwise_fma(nl, nlog2e, tv * s2v);   // (no column term: the record holds 1e30)
            const f32x2 pr = {fast_exp2(ar[0]), fast_exp2(ar[1])};
            const f32x2 pc = {fast_exp2(ac[0]), fast_exp2(ac[1])};
            f32x2 g = __builtin_elementwise_fma(c_colv, pc, c_rowv * pr), gs = __builtin_elementwise_fma(s_colv, pc, s_rowv * pr);
            if (GEN) {
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const int jglob = jt0 + 32 * wjt + 8 * q4 + 4 * h + 2 * u2 + u;
                const bool diag = jglob == lab, valid = ivalid & (jglob < n_cols);
                g[u] -= diag ? c_diag : 0.f;
                gs[u] -= diag ? s_diag : 0.f;
                g[u] = valid ? g[u] : 0.f;
                gs[u] = valid ? gs[u] : 0.f;
              }
            }
            ds2 = __builtin_elementwise_fma(gs, tv, ds2);
            g4[2 * u2] = (bf16_t)g[0];
            g4[2 * u2 + 1] = (bf16_t)g[1];
          }
          *reinterpret_cast<bf16x4*>(gimg + ri * 128 + (((4 * wjt + q4) ^ (ri & 7)) << 4) + 8 * h) = g4;
        }
      };
      const bool plain = rows_full && jt0 + CB_JT <= n_cols && (jt0 + CB_JT <= lab_lo || jt0 >= lab_lo + CB_ROWS);   // wave-uniform
      if (plain) g_step(std::false_type{});
      else g_step(std::true_type{});
    }
    __syncthreads();
    // ---------------- phase D: dX[64 i][128 wave ..] += G[64 i][64 j] Y[64 j][128 wave ..]
    if (!(dbg & 4)) {
      // step q = 4 js + 2 sub + ct: 16 columns j (js), output columns 128 wave + 64 sub + 32 ct ..; the G fragments are read up
      // front, the transposed Y fragments run four steps ahead of their MFMA pairs
      bf16x8 ga[4][2], yt[4];
#pragma unroll
      for (int js = 0; js < 4; ++js)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = 32 * it + r;
          ga[js][it] = *reinterpret_cast<const bf16x8*>(gimg + row * 128 + (((2 * js + h) ^ (row & 7)) << 4));
        }
      auto read_yt = [&](int q) {
        const int js = q >> 2, sub = (q >> 1) & 1, ct = q & 1;
        const char* img = yb + (8 * (js >> 1) + 2 * wave + sub) * CB_SUB + (js & 1) * 2048;
        return tr8(img + troff[0][ct], img + troff[1][ct]);
      };
#pragma unroll
      for (int q = 0; q < 4; ++q) yt[q] = read_yt(q);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int js = q >> 2, kt = q & 3;
        dacc[0][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[js][0], yt[q & 3], dacc[0][kt], 0, 0, 0);
        dacc[1][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[js][1], yt[q & 3], dacc[1][kt], 0, 0, 0);
        if (q + 4 < 16) yt[q & 3] = read_yt(q + 4);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
  }
  // ---- the f32 tile -> this split's slab: dacc[it][kt][e] = dX[i0 + 32 it + (e&3) + 8(e>>2) + 4h][128 wave + 32 kt + r]
  float* sl = p.slab + ((size_t)split * p.r_pad + i0) * CB_KP + 128 * wave + r;
#pragma unroll
  for (int it = 0; it < 2; ++it)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = 32 * it + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (i0 + row < p.r_pad && !(dbg & 16)) sl[(size_t)row * CB_KP + 32 * kt] = dacc[it][kt][e];
      }
  // ---- d/dscale partial of this workgroup
  __syncthreads();
  float v = ds2[0] + ds2[1];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  float* red = reinterpret_cast<float*>(smem);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (tid == 0) p.ds_part[rb * batch.n_split + split] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Forward statistics of the same directions (row shards): per owned row the log-sum-exp partial over one column split and the
// positive logit.  The general tile kernel (clip.hip, gemm_nt_kernel with the statistics epilogue) runs such a shape as 1,024
// independent 128 x 128 tiles, each with its own prologue and epilogue around eight K steps: 28 us for 17 GFLOP at R = 1024 x
// C = 8192.  Here a workgroup owns 64 rows x one column split as in the backward: the X fragments stay in registers, the Y tiles
// stream through the two LDS buffers, and a lane keeps the running (maximum, sum) of ONE row over all the tiles of its split -- the
// online form of the reduction, in the log2 domain, with the row's 16 logits of a tile in its own registers (S^T layout: no
// cross-lane step until the very end).  Column statistics are not made: at W > 1 they come from the other direction's rows through
// the all-reduce.  One partial per (split, row); lse_merge_kernel (clip.hip) merges them as it merges the tile partials.
constexpr int CF_LDS = 2 * CB_YBUF + 1024;   // two Y tiles + the cross-wave exchange of the final partials

__global__ __launch_bounds__(256) void clip_fwd_shard_kernel(const FwdShardBatch batch, const float* __restrict__ scale_ptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ybuf = smem;
  float2* xch = reinterpret_cast<float2*>(smem + 2 * CB_YBUF);   // [2 wit][32 rows]: wave (wit, 1) hands its partial to wave (wit, 0)
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wit = wave & 1, wjt = wave >> 1;
  const int lin = blockIdx.x, xcd = lin & 7, jj_ = lin >> 3;
  const int vunit = xcd + 8 * (jj_ / batch.rows_per_group);
  const int unit = vunit / batch.groups, rb = (vunit % batch.groups) * batch.rows_per_group + jj_ % batch.rows_per_group;
  if (unit >= batch.n_probs * batch.n_split || rb >= batch.row_blocks) return;
  const int split = unit % batch.n_split;
  const FwdShardProb& p = batch.p[unit / batch.n_split];
  const int i0 = rb * CB_ROWS;
  if (i0 >= p.r) return;
  const int n_cols = p.c;
  const bf16_t* y_rows = p.y;
  asm volatile("" : "+s"(y_rows));
  const int c0 = split * batch.cols_per_split;
  const int ntile = max(0, (min(n_cols, c0 + batch.cols_per_split) - c0 + CB_JT - 1) / CB_JT);
  const float sc = *scale_ptr, s2 = sc * 1.4426950408889634f;

  int rowoff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) rowoff[kk] = r * 128 + (((2 * kk + h) ^ cb_swz(r)) << 4);
  const uint32_t smem_addr = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)smem);
  auto issue_piece = [&](const bf16_t* base, int avail, int b, int idx) {   // as in the backward: 16 pieces per wave and tile, rows clamped
    const int jt = idx >> 3, sub = (idx >> 2) & 1, q8 = idx & 3;
    const int s = 2 * wave + sub;
    const int row = 8 * q8 + (lane >> 3);
    const int ch = (lane & 7) ^ cb_swz(row);
    cb_dma16(base + 64 * s, (uint32_t)(min(32 * jt + row, avail) * CB_KP + ch * 8) * 2u,
             smem_addr + b * CB_YBUF + (8 * jt + s) * CB_SUB + q8 * 1024);
  };
  auto issue_rows = [&](const bf16_t* base, int avail, int b) {
#pragma unroll
    for (int idx = 0; idx < 16; ++idx) issue_piece(base, avail, b, idx);
  };
  const int ri = 32 * wit + r, iglob = i0 + ri;
  const int lab = p.label_off + iglob;
  const int lab_lo = p.label_off + i0;

  if (ntile > 0) issue_rows(y_rows + (long)c0 * CB_KP, n_cols - 1 - c0, 0);
  issue_rows(p.x + (long)i0 * CB_KP, p.r - 1 - i0, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bf16x8 xf[32];
#pragma unroll
  for (int ks = 0; ks < 32; ++ks)
    xf[ks] = *reinterpret_cast<const bf16x8*>(ybuf + CB_YBUF + (8 * wit + (ks >> 2)) * CB_SUB + rowoff[ks & 3]);
#pragma unroll
  for (int ks = 0; ks < 32; ++ks) asm volatile("" : "+v"(xf[ks]));

  float m_run = -INFINITY, l_run = 0.f;   // this lane's row: running maximum (log2 domain) and sum of 2^(u - m_run) over its 16 columns per tile
  // (A variant that carried the statistics of tile t between the MFMAs of tile t + 1 measured the same 24 us: with one wave per SIMD
  // the tile time follows the wave's instruction count, not what overlaps what.)
#pragma unroll 1
  for (int t = 0; t < ntile; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t
    __syncthreads();                                   // tile t complete; every wave has finished reading tile t - 1: its buffer is free
    const int tn = min(t + 1, ntile - 1), nb = (t + 1) & 1;
    const bf16_t* ynext = y_rows + (long)(c0 + CB_JT * tn) * CB_KP;
    const int avail_next = n_cols - 1 - (c0 + CB_JT * tn);
    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
    {
      uint32_t ya[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) ya[kk] = smem_addr + (t & 1) * CB_YBUF + 8 * wjt * CB_SUB + rowoff[kk];
      bf16x8 yf[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(yf[ks]) : "v"(ya[ks & 3]), "n"((ks >> 2) * CB_SUB) : "memory");
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) {
        if (ks <= 24) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(yf[ks & 7]) : : "memory");
        else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(yf[ks & 7]) : "n"(31 - ks) : "memory");
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[ks & 7], xf[ks], sacc, 0, 0, 0);
        if (ks + 8 < 32)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(yf[ks & 7]) : "v"(ya[ks & 3]), "n"(((ks + 8) >> 2) * CB_SUB) : "memory");
        if (ks & 1) issue_piece(ynext, avail_next, nb, ks >> 1);
      }
    }
    // ---- element e: logit of row ri against column jt0 + 32 wjt + (e&3) + 8 (e>>2) + 4 h
    const int jt0 = c0 + CB_JT * t;
    const int jb = jt0 + 32 * wjt + 4 * h;
    float u[16];
    const bool edge = jt0 + CB_JT > n_cols;                                            // wave-uniform: the ragged last tile
    const bool has_lab = !(jt0 + CB_JT <= lab_lo || jt0 >= lab_lo + CB_ROWS);          // wave-uniform: a label column in this tile
#pragma unroll
    for (int e = 0; e < 16; ++e) u[e] = sacc[e] * s2;
    if (edge) {
#pragma unroll
      for (int e = 0; e < 16; ++e) u[e] = (jb + (e & 3) + 8 * (e >> 2) < n_cols) ? u[e] : -INFINITY;
    }
    if (has_lab) {
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (jb + (e & 3) + 8 * (e >> 2) == lab && iglob < p.r) p.diag[iglob] = sc * sacc[e];
    }
    float mt = u[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) mt = fmaxf(mt, u[e]);
    const float mn = fmaxf(m_run, mt);
    if (mn > -INFINITY) {   // (a lane whose 16 columns of the ragged tile are all beyond c, with nothing before: nothing to add)
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) sum += fast_exp2(u[e] - mn);
      l_run = fmaf(l_run, fast_exp2(m_run - mn), sum);
      m_run = mn;
    }
  }
  // ---- the row's partial over this split: the two column halves of the lane pair (h), then the two column-half waves (wjt)
  {
    const float mo = __shfl_xor(m_run, 32), lo = __shfl_xor(l_run, 32);
    const float mn = fmaxf(m_run, mo);
    if (mn > -INFINITY) {
      l_run = l_run * fast_exp2(m_run - mn) + lo * fast_exp2(mo - mn);
      m_run = mn;
    }
  }
  __syncthreads();
  if (wjt == 1 && h == 0) xch[32 * wit + r] = make_float2(m_run, l_run);
  __syncthreads();
  if (wjt == 0 && h == 0 && iglob < p.r) {
    const float2 o = xch[32 * wit + r];
    const float mn = fmaxf(m_run, o.x);
    float l = 0.f;
    if (mn > -INFINITY) l = l_run * fast_exp2(m_run - mn) + o.y * fast_exp2(o.x - mn);
    p.part[(size_t)split * p.part_ld + iglob] = make_float2(mn, l);
  }
}

int launch_clip_fwd_shard(const FwdShardBatch& b, const float* scale, hipStream_t st) {
  auto kern = clip_fwd_shard_kernel;
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(kern), 256, CF_LDS, &ks)) return rc;
  FwdShardBatch bb = b;
  const int units = b.n_probs * b.n_split;
  bb.groups = units < 8 ? cdiv(8, units) : 1;
  bb.rows_per_group = cdiv(b.row_blocks, bb.groups);
  const int grid = 8 * cdiv(units * bb.groups, 8) * bb.rows_per_group;
  ProfEvents pe(MMK_K_SIM_STATS);
  hipExtLaunchKernelGGL(kern, dim3(grid), dim3(256), CF_LDS, st, pe.start, pe.stop, 0, bb, scale);
  MMK_LAUNCH_CHECK();
  return 0;
}

// host side: called by clip_backward_impl (clip.hip) for the directions it found eligible
int launch_clip_bwd_fused(const BwdFusedBatch& b, const float* scale, hipStream_t st) {
  auto kern = clip_bwd_fused_kernel;
  KernelSetup ks;
  if (int rc = kernel_setup(reinterpret_cast<const void*>(kern), 256, CB_LDS, &ks)) return rc;
  BwdFusedBatch bb = b;
  bb.dbg = MMK_DBG_ENV("MMK_CB_DBG") ? atoi(MMK_DBG_ENV("MMK_CB_DBG")) : 0;
  const int units = b.n_probs * b.n_split;
  bb.groups = units < 8 ? cdiv(8, units) : 1;
  bb.rows_per_group = cdiv(b.row_blocks, bb.groups);
  const int grid = 8 * cdiv(units * bb.groups, 8) * bb.rows_per_group;
  ProfEvents pe(MMK_K_CLIP_BWD_FUSED);
  hipExtLaunchKernelGGL(kern, dim3(grid), dim3(256), CB_LDS, st, pe.start, pe.stop, 0, bb, scale);
  MMK_LAUNCH_CHECK();
  return 0;
}

}  // namespace mmk
