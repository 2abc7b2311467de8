// CLIP-style InfoNCE over a similarity matrix, for gfx950 (MI355X).
//
// Replaces the ATen sequence of mmlearn/modules/losses/contrastive.py:134-144,327-340
// (_safe_matmul -> logit_scale * -> F.cross_entropy x2 and their autograd) with:
//
//   forward : sim_stats   T[j][i] = Y[j].X[i] on MFMA, fused online row-LSE over j per owned row i
//             lse_reduce  merges the per-column-tile (max,sum) partials, sums (lse_i - S_i,label)
//   backward: sim_grad    recomputes the tile, writes G = c_row*P_row + c_col*P_col - c_diag*delta
//             grad_gemm   dX = G @ Y   (split-K, f32 slabs)
//             finalize    sums slabs, scales, (L2-norm backward), scatters to the user gradient
//
// One GEMM main loop serves all three MFMA kernels:  T[m][n] = sum_k P[m][k] * Q[n][k]
// (both operands K-contiguous).  P rows land on the accumulator REGISTERS, Q rows on the
// LANES (MFMA A = P, B = Q^T), so the reduction over m that the row-LSE needs is lane-local.
//
// Data layout: operands are [rows][k_pad] with k_pad a multiple of one 128-byte LDS row
// (64 bf16 / 32 f32).  LDS tiles are [rows][128 B], 16-byte chunks XOR-swizzled with
// (row>>1)&7 so that the 16-lane groups of ds_read_b128 hit 16 distinct 16-B slots.
#include <hip/hip_ext.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "clip_internal.h"

extern "C" int mmk_wgrad_plan(int64_t M, int N, int K, int* splits_out, int64_t* ws_floats_out);
extern "C" int mmk_wgrad_partial(const void* dy, const void* x, float* ws, int64_t M, int N, int K, int64_t ldy, int64_t ldx, int32_t* splits_out,
                                 int32_t* n_pad_out, int32_t* k_pad_out, void* stream);

namespace mmk {


// ---------------------------------------------------------------- MMA atoms
// One "macro k-step" consumes 32 bytes of K per operand row: lane (r = lane&31, h = lane>>5)
// reads the 16-byte chunk (2*kk + h) of row r.  bf16: one v_mfma_f32_32x32x16_bf16 (k = 8h+j).
// f32: four v_mfma_f32_32x32x2_f32 (MFMA t pairs k = 4h+t of both halves) -- exact f32.
template <typename T>
struct Atom;
template <>
struct Atom<bf16_t> {
  static constexpr int BK = 64;
  typedef bf16x8 Frag;
  static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Atom<float> {
  static constexpr int BK = 32;
  typedef f32x4 Frag;
  static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
  }
};

enum { EPI_STATS = 0, EPI_GRAD = 1, EPI_PLAIN = 2, EPI_ALIGN_STATS = 3, EPI_ALIGN_GRAD = 4 };

struct Prob {
  const char* P;  // rows -> accumulator registers (MFMA A)
  const char* Q;  // rows -> lanes                 (MFMA B)
  int M, N;       // valid rows of P / Q
  int K;          // elements, multiple of Atom::BK
  int ldp, ldq;   // elements
  int tiles_m, tiles_n;
  // EPI_STATS
  float2* part;   // [tiles_m][part_ld = N]  (column-tile major: coalesced for the merge)
  int part_ld;
  float* diag;    // [N]
  int label_off;
  float2* mpart;  // EPI_STATS, mirrored direction: [tiles_n * 2][mpart_ld = M] column (ref2, sum) partials, or null
  int mpart_ld;
  const float* qn;      // [>= N] L2 norms of the Q rows as packed (what the MFMA multiplies), or null: no bounded fast path
  const float* pn;      // [>= M] the same for the P rows
  char* GT;       // EPI_GRAD: optional transposed G [>= tiles_m*BM][ldgt]
  int ldgt;
  // EPI_GRAD
  const float* lse_row;  // [N]
  const float* lse_col;  // [M]
  char* G;               // [>= tiles_n*BN][ldg]
  int ldg;
  float c_row, c_col, c_diag, s_row, s_col, s_diag;
  float* ds_part;        // [tiles_m * tiles_n]
  // EPI_ALIGN_*: Q row i is row (label_off + i) of the concatenated feature matrix P (M rows)
  const int* hmax;       // [M]: positives of row r are the columns [r, hmax[r])
  // EPI_PLAIN (split-K)
  float* slab;           // [n_split][slab_rows][slab_ld]
  int slab_ld;
  long slab_split_stride;
  int k_per_split;
};
struct ProbBatch {
  Prob p[MAX_PROBS];
  int n_split;
  int n_probs;
  int unit_map;           // EPI_PLAIN: 1 = every (problem, K split) unit lives on ONE XCD (see the tile decode of gemm_nt_kernel)
  int dbg;                // -DMMK_DEBUG_SWITCHES builds only (MMK_SIM_DBG; wrong results): 1 = no row statistics, 2 = no column
                          // statistics, 4 = no MFMAs, 16 = no main loop, 32 = return at once; 8 = never take the bounded fast path.
                          // The shipped build ignores the field; the exact path is selected by passing no row norms.
};

// ------------------------------------------------------------------ main loop
// LOADER 0: global -> VGPR -> ds_write_b128 staging (2 LDS stages).
// LOADER 1: LDS-DMA (global_load_lds_dwordx4): each wave-instruction lands 1 KiB = 8 tile rows linearly in
//           LDS; the XOR swizzle is applied to the per-lane SOURCE chunk instead (the read side is unchanged).
//           NSTAGE LDS stages, counted s_waitcnt vmcnt(N) + raw s_barrier so prefetches stay in flight.
enum { LOADER_REG = 0, LOADER_DMA = 1 };

__device__ __forceinline__ void glds16(const char* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WM = waves along the P rows (m); two waves along the Q rows (n).  WM = 2: four waves, two workgroups per CU (128 x 128 / 64 x 64
// tiles); WM = 4: eight waves on a 256 x 128 tile, one workgroup per CU with three stages (a quarter fewer bytes into LDS per
// output, more of them in flight).  A wave's piece is (BM / WM) x (BN / 2) in both cases.
template <typename T, int BM, int BN, int EPI, int LOADER, int NSTAGE, int WM = 2>
__global__ __launch_bounds__(128 * WM, WM == 2 ? 2 : 1) void gemm_nt_kernel(const ProbBatch batch, const float* __restrict__ scale_ptr) {
  typedef Atom<T> A;
  typedef typename A::Frag Frag;
  constexpr int NWAVES = 2 * WM, NTHREADS = 64 * NWAVES;
  constexpr int MT = BM / (32 * WM), NT = BN / 64;   // 32x32 MFMA tiles per wave per dim (WM x 2 waves)
  constexpr int ROWS = BM + BN;
  constexpr int CPT = ROWS * 8 / NTHREADS;                // 16-byte chunks per thread per stage (= DMA instr per wave)
  constexpr int STAGE_BYTES = ROWS * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NSTAGE * STAGE_BYTES (dynamic: may exceed 64 KiB)

  int zsplit = (EPI == EPI_PLAIN) ? (blockIdx.z % batch.n_split) : 0;
  int zprob = (EPI == EPI_PLAIN) ? (blockIdx.z / batch.n_split) : blockIdx.z;
  int raw_tile = blockIdx.x;
  // Row-sharded gradient GEMM (few G row tiles, many K splits: R = 1024 x C = 8192 is 32 tiles x 8 splits x 2 directions): a unit =
  // (problem, split) reads ITS K range of G and of Y^T and nothing else, so all tiles of a unit go to one XCD -- G and Y^T are then
  // fetched once.  (The tile-range map below keeps a G row tile in one L2 but has all eight XCDs read every Y^T range: 128 of the
  // 180 MB this launch fetched for 48 MB of operands, profiles/r04_pmc_traffic_shard.json.)  Workgroups are dealt to the XCDs
  // round-robin in dispatch order (x fastest, then z); the host sets unit_map when the unit count is a multiple of 8.
  const bool unit_map = EPI == EPI_PLAIN && batch.unit_map;
  if (unit_map) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.z, xcd = lin & 7, j = lin >> 3;
    const int unit = xcd + 8 * (j / (int)gridDim.x);
    raw_tile = j % (int)gridDim.x;
    zsplit = unit % batch.n_split;
    zprob = unit / batch.n_split;
  }
  const Prob& p = batch.p[zprob];
  const int n_tiles = p.tiles_m * p.tiles_n;
  if (raw_tile >= n_tiles) return;
  const int dbg = kDebugSwitches ? batch.dbg : 0;   // folded to 0 in the shipped build: no ablation branch survives
  // XCD-aware tile order (speed only, any placement is correct): workgroups are dealt round-robin over the 8 XCDs,
  // so XCD x can be given the contiguous tile range [base_x, base_x + cnt_x): neighbours in (tn, tm) order then share
  // the Q row tile in one L2 instead of it being fetched by all eight (rocprofv3 FETCH_SIZE, N = 8192: 1082 -> 392 MB).
  int tile = raw_tile;
  if (EPI == EPI_PLAIN && !unit_map) {
    // gradient GEMM: the k_pad/BM blocks that share a G row tile must meet in one L2 (G does not fit any cache).
    // The similarity kernels keep the plain order: there each XCD sees every 8th P tile, a 1/8 slice of P that
    // stays L2-resident while Q streams through once (measured: 142 MB fetched vs 1040 MB with the remap).
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int q = n_tiles >> 3, rr = n_tiles & 7;
    tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  const int tm = tile % p.tiles_m, tn = tile / p.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  int k_begin = 0, k_end = p.K;
  if (EPI == EPI_PLAIN) {
    k_begin = zsplit * p.k_per_split;
    k_end = min(p.K, k_begin + p.k_per_split);
  }
  if (dbg & 32) return;                                        // timing: launch + dispatch only
  const int nk = (dbg & 16) ? 0 : (k_end - k_begin) / A::BK;   // timing: no main loop (launch + epilogue)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM, r = lane & 31, h = lane >> 5;

  // ---- staging assignment.  REG: chunk c = tid + 256*u (8 chunks per row), swizzled LDS destination.
  //      DMA: wave-instruction u of wave w covers tile rows 8*(w*CPT+u) .. +7; lane L lands at LDS row
  //      8g + (L>>3), slot L&7 and therefore fetches source chunk (L&7) ^ swizzle(row).
  const char* gsrc[CPT];
  int lds_off[CPT];
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    int row, ch;
    if (LOADER == LOADER_REG) {
      const int c = tid + NTHREADS * u;
      row = c >> 3;
      ch = c & 7;
      lds_off[u] = row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
    } else {
      const int g = wave * CPT + u;
      row = 8 * g + (lane >> 3);
      ch = (lane & 7) ^ ((row >> 1) & 7);
      lds_off[u] = g * 1024;  // wave-uniform base of this instruction's 1 KiB
    }
    const char* base;
    if (row < BM) {
      const int g = min(m0 + row, p.M - 1);
      base = p.P + ((size_t)g * p.ldp + k_begin) * sizeof(T);
    } else {
      const int g = min(n0 + (row - BM), p.N - 1);
      base = p.Q + ((size_t)g * p.ldq + k_begin) * sizeof(T);
    }
    gsrc[u] = base + ch * 16;
  }

  // EPI_STATS / EPI_GRAD: L2 norm of "this thread's" operand row (threads 0 .. BN-1: the tile's Q rows, BN .. BN+BM-1: its P rows),
  // fetched now, reduced to the tile's bound after the main loop
  float my_nrm = 0.f;
  if ((EPI == EPI_STATS || EPI == EPI_GRAD) && p.qn != nullptr) {
    if (tid < BN) my_nrm = (n0 + tid < p.N) ? p.qn[n0 + tid] : 0.f;
    else if (tid < BN + BM) my_nrm = (m0 + tid - BN < p.M) ? p.pn[m0 + tid - BN] : 0.f;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // fragment read offsets (row-dependent swizzle is constant over k)
  int p_off[MT], q_off[NT], p_sw[MT], q_sw[NT];
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    const int row = wm * (BM / WM) + a * 32 + r;
    p_off[a] = row * 128;
    p_sw[a] = (row >> 1) & 7;
  }
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    const int row = BM + wn * (BN / 2) + b * 32 + r;
    q_off[b] = row * 128;
    q_sw[b] = (row >> 1) & 7;
  }

  auto compute_stage = [&](const char* cur) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      Frag fa[MT], fb[NT];
      const int ch = 2 * kk + h;
#pragma unroll
      for (int a = 0; a < MT; ++a) fa[a] = *reinterpret_cast<const Frag*>(cur + p_off[a] + ((ch ^ p_sw[a]) << 4));
#pragma unroll
      for (int b = 0; b < NT; ++b) fb[b] = *reinterpret_cast<const Frag*>(cur + q_off[b] + ((ch ^ q_sw[b]) << 4));
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) A::mma(fa[a], fb[b], acc[a][b]);
    }
  };

  if (LOADER == LOADER_REG) {
    uint4 stage[CPT];
    if (nk > 0) {
#pragma unroll
      for (int u = 0; u < CPT; ++u) stage[u] = *reinterpret_cast<const uint4*>(gsrc[u]);
#pragma unroll
      for (int u = 0; u < CPT; ++u) *reinterpret_cast<uint4*>(smem + lds_off[u]) = stage[u];
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const char* cur = smem + (kt & 1) * STAGE_BYTES;
      const bool more = kt + 1 < nk;
      if (more) {
#pragma unroll
        for (int u = 0; u < CPT; ++u) stage[u] = *reinterpret_cast<const uint4*>(gsrc[u] + (size_t)(kt + 1) * 128);
      }
      compute_stage(cur);
      if (more) {
        char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
#pragma unroll
        for (int u = 0; u < CPT; ++u) *reinterpret_cast<uint4*>(nxt + lds_off[u]) = stage[u];
      }
      __syncthreads();
    }
  } else {
    // prologue: NSTAGE-1 stages in flight (empty ones are still "issued" as no-ops by skipping: counts below
    // assume exactly CPT DMA instructions per issued stage, so issue only real stages and clamp the wait)
    auto issue = [&](int kt) {
      char* dst = smem + (kt % NSTAGE) * STAGE_BYTES;
#pragma unroll
      for (int u = 0; u < CPT; ++u) glds16(gsrc[u] + (size_t)kt * 128, dst + lds_off[u]);
    };
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st)
      if (st < nk) issue(st);
    for (int kt = 0; kt < nk; ++kt) {
      // stage kt has landed when at most the younger issued stages remain outstanding
      const int younger = min(nk - 1, kt + NSTAGE - 2) - kt;  // stages issued after kt so far
      if (younger >= 2) wait_vmcnt<2 * CPT>();
      else if (younger == 1) wait_vmcnt<CPT>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();  // every wave's share of stage kt is in LDS; stage kt-1 fully consumed
      if (kt + NSTAGE - 1 < nk) issue(kt + NSTAGE - 1);  // refill the buffer that stage kt-1 just released
      if (!(dbg & 4)) compute_stage(smem + (kt % NSTAGE) * STAGE_BYTES);
    }
    wait_vmcnt<0>();
    __syncthreads();
  }

  // Bound of the tile's logits in the log2 domain from the operand norms (Cauchy-Schwarz): |u| <= R for every element.
  // Returns 0 when the norms are unknown.  Block-uniform; contains a barrier (call it from uniform control flow only).
  auto tile_bound = [&](float s2) -> float {
    float* wmax = reinterpret_cast<float*>(smem + WM * BN * 8);       // [NWAVES] (behind the [WM][BN] float2 area of EPI_STATS)
    float v = my_nrm;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    if (lane == 0) wmax[wave] = v;
    __syncthreads();
    float nq = 0.f, np = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) {
      if (w * 64 < BN) nq = fmaxf(nq, wmax[w]);
      else if (w * 64 < BN + BM) np = fmaxf(np, wmax[w]);
    }
    __syncthreads();   // wmax may be overwritten by the caller's LDS use
    return fabsf(s2) * nq * np * 1.0001f;   // margin: f32 accumulation error of the dot product
  };

  // -------------------------------------------------------------- epilogues
  // element (a, b, e): m = m0 + wm*BM/2 + a*32 + (e&3) + 8*(e>>2) + 4*h ; n = n0 + wn*BN/2 + b*32 + r
  constexpr float LOG2E = 1.4426950408889634f;
  if (EPI == EPI_STATS) {
    // All softmax arithmetic runs in the log2 domain (u = s*log2e*t): one v_fma + v_exp per element.  A tile leaves one
    // partial (ref2, sum) per owned row -- sum of 2^(u - ref2) over the tile's columns -- and, for a mirrored direction,
    // one per column and wn half; lse_merge_kernel combines them.
    //   FAST path (interior tile, norms known, 2 R <= 96): by Cauchy-Schwarz every |u| of the tile is <= R = |s2| max|q| max|p|,
    //   so ref2 = R for the whole tile: no maximum search, and ONE exponential per element serves the row sum and the
    //   column sum (each term is >= 2^-96: nothing flushes, the result is exact to rounding).
    //   EXACT path (edge tiles, unknown or large norms): per-row / per-column maxima as references.
    const float s = *scale_ptr;
    const float s2 = s * LOG2E;
    const bool interior = (m0 + BM <= p.M);
    const bool has_diag = (m0 < p.label_off + n0 + BN) && (m0 + BM > p.label_off + n0);
    float2* red = reinterpret_cast<float2*>(smem);                    // [2][BN]
    bool fast = false;
    float R = 0.f;
    if (p.qn != nullptr && interior && (n0 + BN <= p.N) && !(dbg & 8)) {
      R = tile_bound(s2);
      fast = (2.f * R <= 96.f);
    }
    if (has_diag) {
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int i = n0 + wn * (BN / 2) + b * 32 + r;
        const int lab = p.label_off + i;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int j = m0 + wm * (BM / WM) + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (j == lab && i < p.N && j < p.M) p.diag[i] = s * acc[a][b][e];
          }
      }
    }
    const bool want_col = p.mpart != nullptr && !(dbg & 2);
    constexpr int CNT = 16 * MT;
    if (fast) {
      float csum[CNT];
#pragma unroll
      for (int c = 0; c < CNT; ++c) csum[c] = 0.f;
      if (!(dbg & 1)) {
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          float rs = 0.f;
#pragma unroll
          for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const float pv = fast_exp2(fmaf(acc[a][b][e], s2, -R));
              rs += pv;
              csum[a * 16 + e] += pv;
            }
          rs += __shfl_xor(rs, 32);
          if (h == 0) red[wm * BN + wn * (BN / 2) + b * 32 + r] = make_float2(R, rs);
        }
      }
      __syncthreads();
      if (tid < BN) {
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) l += red[w * BN + tid].y;
        p.part[(size_t)tm * p.part_ld + n0 + tid] = make_float2(R, l);
      }
      if (want_col) {
        const float cs = half_wave_transpose_reduce<CNT, false>(csum, lane);   // lane L: column L & (CNT - 1) of half h
        const int cidx = lane & (CNT - 1);
        const int j = m0 + wm * (BM / WM) + (cidx >> 4) * 32 + (cidx & 3) + 8 * ((cidx & 15) >> 2) + 4 * h;
        if (CNT == 32 || (lane & 16) == 0) p.mpart[(size_t)(tn * 2 + wn) * p.mpart_ld + j] = make_float2(R, cs);
      }
    } else {
      if (!(dbg & 1))
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int nl = wn * (BN / 2) + b * 32 + r;
        float sum = 0.f, m2;
        if (interior && s2 >= 0.f) {  // every column valid, max of s*t is at max of t
          float vmax = -INFINITY;
#pragma unroll
          for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) vmax = fmaxf(vmax, acc[a][b][e]);
          vmax = fmaxf(vmax, __shfl_xor(vmax, 32));
          m2 = vmax * s2;
#pragma unroll
          for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += fast_exp2(fmaf(acc[a][b][e], s2, -m2));
        } else {  // edge tiles / negative scale: masked extremum of u = s2*t, masked sum
          float umax = -INFINITY;   // (the accumulators stay untouched: the mirrored column statistics read them again)
#pragma unroll
          for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int j = m0 + wm * (BM / WM) + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
              const float u = (j < p.M) ? acc[a][b][e] * s2 : -INFINITY;
              umax = fmaxf(umax, u);
            }
          umax = fmaxf(umax, __shfl_xor(umax, 32));
          m2 = umax;
          if (umax > -INFINITY) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const int j = m0 + wm * (BM / WM) + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                sum += (j < p.M) ? fast_exp2(acc[a][b][e] * s2 - umax) : 0.f;
              }
          }
        }
        sum += __shfl_xor(sum, 32);
        if (h == 0) red[wm * BN + nl] = make_float2(m2, sum);
      }
      __syncthreads();
      if (tid < BN) {
        float mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < WM; ++w) mx = fmaxf(mx, red[w * BN + tid].x);
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          const float2 x = red[w * BN + tid];
          if (x.x > -INFINITY) l += x.y * fast_exp2(x.x - mx);
        }
        const int i = n0 + tid;
        if (i < p.N) p.part[(size_t)tm * p.part_ld + i] = make_float2(mx, l);  // log2 domain
      }
      if (want_col) {
        // Mirrored direction (W = 1: logits_per_b = logits_per_a^T): the SAME tile gives, per column j (a P row, on the
        // accumulator registers), the (max2, sum) over this wave's 32 * NT rows i (the lanes).  Per lane first over its NT
        // rows, then a transpose-reduce over the 32 lanes of the half-wave: max pass, maxima handed back through a
        // wave-private LDS line, exponentials against the column maximum, sum pass.  One partial per (row tile, wn half).
        __syncthreads();   // the row partials in `red` have been consumed
        float* cmax_lds = reinterpret_cast<float*>(smem) + wave * 64;   // [h][CNT] per wave (CNT <= 32)
        float vmax[CNT];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float m = -INFINITY;
#pragma unroll
            for (int b = 0; b < NT; ++b) {
              const int i = n0 + wn * (BN / 2) + b * 32 + r;
              const float u = (i < p.N) ? acc[a][b][e] * s2 : -INFINITY;
              m = fmaxf(m, u);
            }
            vmax[a * 16 + e] = m;
          }
        const float cm = half_wave_transpose_reduce<CNT, true>(vmax, lane);   // lane L: column L & (CNT - 1) of half h
        cmax_lds[h * 32 + (lane & (CNT - 1))] = cm;   // CNT = 16: two lanes write the same value
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float vsum[CNT];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float mref = cmax_lds[h * 32 + a * 16 + e];   // broadcast read
            float t = 0.f;
            if (mref > -INFINITY) {
#pragma unroll
              for (int b = 0; b < NT; ++b) {
                const int i = n0 + wn * (BN / 2) + b * 32 + r;
                if (i < p.N) t += fast_exp2(fmaf(acc[a][b][e], s2, -mref));
              }
            }
            vsum[a * 16 + e] = t;
          }
        const float cs = half_wave_transpose_reduce<CNT, false>(vsum, lane);
        const int cidx = lane & (CNT - 1);
        const int j = m0 + wm * (BM / WM) + (cidx >> 4) * 32 + (cidx & 3) + 8 * ((cidx & 15) >> 2) + 4 * h;
        if (j < p.M && (CNT == 32 || (lane & 16) == 0)) p.mpart[(size_t)(tn * 2 + wn) * p.mpart_ld + j] = make_float2(cm, cs);
      }
    }
  } else if (EPI == EPI_GRAD) {
    const float s = *scale_ptr;
    const float s2 = s * LOG2E;
    const bool use_col = (p.c_col != 0.f) || (p.s_col != 0.f);
    const bool use_ds = (p.s_row != 0.f) || (p.s_col != 0.f) || (p.s_diag != 0.f);
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);
    const bool has_diag = (m0 < p.label_off + n0 + BN) && (m0 + BM > p.label_off + n0);
    float ds_acc = 0.f;
    // FAST path (interior tile, operand norms known, 2 R <= 96; see EPI_STATS): ONE exponential per element,
    // p0 = 2^(u - R); the row / column softmax values are p0 * 2^(R - lse_row_i) and p0 * 2^(R - lse_col_j), whose
    // factors are per row / per column.  Both lse are >= -R, so the factors stay below 2^96.
    bool fast = false;
    float R = 0.f;
    if (p.qn != nullptr && interior && !(dbg & 8)) {
      R = tile_bound(s2);
      fast = (2.f * R <= 96.f);
    }
    const bool same_s = p.s_row == p.c_row && p.s_col == p.c_col && p.s_diag == p.c_diag;
    float ccf[MT][4][4], scf[MT][4][4];   // c_col * 2^(R - lse_col_j), s_col * ... for this lane's 16 MT columns
    if (fast) {
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int jb = m0 + wm * (BM / WM) + a * 32 + 8 * q + 4 * h;
          float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (use_col) t4 = *reinterpret_cast<const float4*>(p.lse_col + jb);
          const float f0 = use_col ? fast_exp2(R - t4.x * LOG2E) : 0.f, f1 = use_col ? fast_exp2(R - t4.y * LOG2E) : 0.f;
          const float f2 = use_col ? fast_exp2(R - t4.z * LOG2E) : 0.f, f3 = use_col ? fast_exp2(R - t4.w * LOG2E) : 0.f;
          ccf[a][q][0] = p.c_col * f0; ccf[a][q][1] = p.c_col * f1; ccf[a][q][2] = p.c_col * f2; ccf[a][q][3] = p.c_col * f3;
          scf[a][q][0] = p.s_col * f0; scf[a][q][1] = p.s_col * f1; scf[a][q][2] = p.s_col * f2; scf[a][q][3] = p.s_col * f3;
        }
    }
    // G leaves through a wave-private LDS staging tile so that global stores are whole row segments
    // (COLS*sizeof(T) = 128/256 B contiguous per row) instead of 8/16-byte pieces scattered over 32 rows.
    constexpr int COLS = BM / WM;                   // j (columns of G) per wave
    constexpr int PB = 4 * (int)sizeof(T);         // bytes of one 4-element piece
    constexpr int RB = COLS * (int)sizeof(T);      // bytes of one staged row
    constexpr int STRIDE = RB + PB;                // padded: conflict-free piece writes, aligned piece reads
    constexpr int LPR = COLS / 4;                  // lanes per row on read-back
    constexpr int RPI = 64 / LPR;                  // rows per read-back instruction
    typedef typename std::conditional<sizeof(T) == 2, uint2, uint4>::type Piece;
    char* stg = smem + wave * (32 * STRIDE);
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int i = n0 + wn * (BN / 2) + b * 32 + r;
      const bool iv = i < p.N;
      const float lr2 = (iv ? p.lse_row[i] : 0.f) * LOG2E;
      const int lab = p.label_off + i;
      const float fr = fast ? fast_exp2(R - lr2) : 0.f;
      const float crf = p.c_row * fr, srf = p.s_row * fr;
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int jl = a * 32 + 8 * q + 4 * h;  // column inside the wave's strip
          const int jb = m0 + wm * (BM / WM) + jl;
          float g4[4];
          if (fast) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float t = acc[a][b][4 * q + e];
              const float p0 = fast_exp2(fmaf(t, s2, -R));
              float g = p0 * (crf + ccf[a][q][e]);
              float gs = same_s ? g : p0 * (srf + scf[a][q][e]);
              if (has_diag && (jb + e == lab)) {
                g -= p.c_diag;
                gs -= p.s_diag;
              }
              if (use_ds) ds_acc = fmaf(gs, t, ds_acc);
              g4[e] = g;
              acc[a][b][4 * q + e] = g;
            }
            Vec4<T>::store(reinterpret_cast<T*>(stg + r * STRIDE) + jl, make_float4(g4[0], g4[1], g4[2], g4[3]));
            continue;
          }
          float lc2[4] = {0.f, 0.f, 0.f, 0.f};
          if (use_col) {
            if (jb + 3 < p.M) {
              const float4 t4 = *reinterpret_cast<const float4*>(p.lse_col + jb);
              lc2[0] = t4.x * LOG2E; lc2[1] = t4.y * LOG2E; lc2[2] = t4.z * LOG2E; lc2[3] = t4.w * LOG2E;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (jb + e < p.M) lc2[e] = p.lse_col[jb + e] * LOG2E;
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = acc[a][b][4 * q + e];
            const float pr = fast_exp2(fmaf(t, s2, -lr2));
            const float pc = use_col ? fast_exp2(fmaf(t, s2, -lc2[e])) : 0.f;
            float g = p.c_row * pr + p.c_col * pc;
            float gs = p.s_row * pr + p.s_col * pc;
            if (has_diag && (jb + e == lab)) {
              g -= p.c_diag;
              gs -= p.s_diag;
            }
            if (!interior && !(iv && (jb + e < p.M))) {
              g = 0.f;
              gs = 0.f;
            }
            if (use_ds) ds_acc = fmaf(gs, t, ds_acc);
            g4[e] = g;
            acc[a][b][4 * q + e] = g;   // kept for the transposed copy below (the logit is no longer needed)
          }
          Vec4<T>::store(reinterpret_cast<T*>(stg + r * STRIDE) + jl, make_float4(g4[0], g4[1], g4[2], g4[3]));
        }
      // read the 32 staged rows back as whole rows and stream them out
#pragma unroll
      for (int t = 0; t < 32 / RPI; ++t) {
        const int row = t * RPI + lane / LPR, c4 = lane % LPR;
        const Piece v = *reinterpret_cast<const Piece*>(stg + row * STRIDE + c4 * PB);
        const int gi = n0 + wn * (BN / 2) + b * 32 + row;
        const int gj = m0 + wm * (BM / WM) + c4 * 4;
        *reinterpret_cast<Piece*>(reinterpret_cast<T*>(p.G) + (size_t)gi * p.ldg + gj) = v;
      }
      if (p.GT != nullptr) {
        // G^T = the mirrored direction's G: the same values staged transposed ([COLS columns j][32 rows i], the wave's
        // staging area again: its LDS operations are in order) and streamed out as whole 32-element row segments
        T* st = reinterpret_cast<T*>(stg);
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int e = 0; e < 16; ++e) st[(a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = from_f32<T>(acc[a][b][e]);
        constexpr int TLPR = 32 * (int)sizeof(T) / 16;   // lanes per staged row (16-byte pieces)
        constexpr int TRPI = 64 / TLPR;                  // rows per read-back instruction
#pragma unroll
        for (int t = 0; t < COLS / TRPI; ++t) {
          const int row = t * TRPI + lane / TLPR, c16 = lane % TLPR;
          const uint4 v = *reinterpret_cast<const uint4*>(stg + row * 32 * (int)sizeof(T) + c16 * 16);
          const int gj = m0 + wm * (BM / WM) + row;
          const int gi = n0 + wn * (BN / 2) + b * 32 + c16 * (16 / (int)sizeof(T));
          *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.GT) + (size_t)gj * p.ldgt + gi) = v;
        }
      }
    }
    __syncthreads();  // staging tiles are dead; reuse LDS for the d/dscale reduction
    ds_acc = wave_sum(ds_acc);
    float* red = reinterpret_cast<float*>(smem);
    if (lane == 0) red[wave] = ds_acc;
    __syncthreads();
    if (tid == 0) {
      float t_ = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) t_ += red[w];
      p.ds_part[tile] = t_;
    }
  } else if (EPI == EPI_ALIGN_STATS) {
    // modality-alignment BCE (contrastive.py:387-413): per row r sums of BCE-with-logits over its positive
    // columns [r, hmax[r]) and over the rest; pointwise, so tile partials simply add.
    const float s = *scale_ptr;
    float2* red = reinterpret_cast<float2*>(smem);  // [2][BN]
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int nl = wn * (BN / 2) + b * 32 + r;
      const int i = n0 + nl;
      const int row = p.label_off + i;
      const int hm = (i < p.N) ? p.hmax[row] : 0;
      float pos = 0.f, neg = 0.f;
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int c = m0 + wm * (BM / WM) + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const float v = s * acc[a][b][e];
          const bool y = (c >= row) && (c < hm);
          // max(v,0) - v*y + log1p(exp(-|v|))
          const float bce = fmaxf(v, 0.f) - (y ? v : 0.f) + log1pf(__expf(-fabsf(v)));
          if (c < p.M) {
            if (y) pos += bce; else neg += bce;
          }
        }
      pos += __shfl_xor(pos, 32);
      neg += __shfl_xor(neg, 32);
      if (h == 0) red[wm * BN + nl] = make_float2(pos, neg);
    }
    __syncthreads();
    if (tid < BN) {
      float2 t2 = make_float2(0.f, 0.f);
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        t2.x += red[w * BN + tid].x;
        t2.y += red[w * BN + tid].y;
      }
      const int i = n0 + tid;
      if (i < p.N) p.part[(size_t)tm * p.part_ld + i] = make_float2(t2.x, t2.y);
    }
  } else if (EPI == EPI_ALIGN_GRAD) {
    // d/dlogits of the alignment loss, symmetrised because logits = s F F^T:  G[r][c] = a_rc + a_cr with
    // a_rc = (sigmoid(v) - y_rc) * (y_rc ? 1/npos_r : 1/nneg_r);  d/dscale uses a_rc only (each (r,c) once).
    const float s = *scale_ptr;
    float ds_acc = 0.f;
    constexpr int COLS = BM / WM;
    constexpr int PB = 4 * (int)sizeof(T);
    constexpr int RB = COLS * (int)sizeof(T);
    constexpr int STRIDE = RB + PB;
    constexpr int LPR = COLS / 4;
    constexpr int RPI = 64 / LPR;
    typedef typename std::conditional<sizeof(T) == 2, uint2, uint4>::type Piece;
    char* stg = smem + wave * (32 * STRIDE);
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int i = n0 + wn * (BN / 2) + b * 32 + r;
      const bool iv = i < p.N;
      const int row = p.label_off + i;
      const int hm_r = iv ? p.hmax[row] : row + 1;
      const float ipos_r = 1.f / (float)(hm_r - row);
      const float ineg_r = 1.f / (float)(p.M - (hm_r - row));
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int jl = a * 32 + 8 * q + 4 * h;
          const int cb = m0 + wm * (BM / WM) + jl;
          float g4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = cb + e;
            const bool cv = c < p.M;
            const int hm_c = cv ? p.hmax[c] : c + 1;
            const float t = acc[a][b][4 * q + e];
            const float v = s * t;
            const float sig = 1.f / (1.f + __expf(-v));
            const bool y_rc = (c >= row) && (c < hm_r);
            const bool y_cr = (row >= c) && (row < hm_c);
            const float w_rc = y_rc ? ipos_r : ineg_r;
            const float w_cr = y_cr ? 1.f / (float)(hm_c - c) : 1.f / (float)(p.M - (hm_c - c));
            const float a_rc = (sig - (y_rc ? 1.f : 0.f)) * w_rc;
            const float a_cr = (sig - (y_cr ? 1.f : 0.f)) * w_cr;
            float g = a_rc + a_cr;
            if (!(iv && cv)) g = 0.f;
            else ds_acc = fmaf(a_rc, t, ds_acc);
            g4[e] = g;
          }
          Vec4<T>::store(reinterpret_cast<T*>(stg + r * STRIDE) + jl, make_float4(g4[0], g4[1], g4[2], g4[3]));
        }
#pragma unroll
      for (int t = 0; t < 32 / RPI; ++t) {
        const int srow = t * RPI + lane / LPR, c4 = lane % LPR;
        const Piece v = *reinterpret_cast<const Piece*>(stg + srow * STRIDE + c4 * PB);
        const int gi = n0 + wn * (BN / 2) + b * 32 + srow;
        const int gj = m0 + wm * (BM / WM) + c4 * 4;
        *reinterpret_cast<Piece*>(reinterpret_cast<T*>(p.G) + (size_t)gi * p.ldg + gj) = v;
      }
    }
    __syncthreads();
    ds_acc = wave_sum(ds_acc);
    float* red = reinterpret_cast<float*>(smem);
    if (lane == 0) red[wave] = ds_acc;
    __syncthreads();
    if (tid == 0) {
      float t_ = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) t_ += red[w];
      p.ds_part[tile] = t_;
    }
  } else {  // EPI_PLAIN: slab[split][n][m] = acc
    float* slab = p.slab + (size_t)zsplit * p.slab_split_stride;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int i = n0 + wn * (BN / 2) + b * 32 + r;
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int mb = m0 + wm * (BM / WM) + a * 32 + 8 * q + 4 * h;
          if (i < p.N && mb < p.M) {
            float4 v = make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
            *reinterpret_cast<float4*>(slab + (size_t)i * p.slab_ld + mb) = v;
          }
        }
    }
  }
}

// ------------------------------------------------------------------ merge of the tile partials
// lse[i] = log-sum-exp over the tile partials of row i (coalesced: partials are tile-major), loss_part[block] = sum over the
// block's 64 rows of (lse_i - diag_i).  A workgroup takes 64 rows; its four waves each merge a quarter of the partials of
// those rows (online: one pass), wave 0 combines the quarters.  Optionally the weighted loss value
// (contrastive.py:134-144,160) in the same launch: every workgroup publishes its weighted block sum with a device-scope
// store, drains it and draws a ticket; the workgroup that draws the last ticket adds all block sums in a fixed order
// (deterministic) and re-arms the counter, which is zero on entry and zero on exit.
struct MergeProb {
  const float2* part;
  int part_ld, n_part, N;
  const float* diag;
  float* lse;
  float* loss_part;   // [cdiv(N, 64)]
  float w;
};
struct MergeBatch {
  MergeProb p[2 * MAX_PROBS];   // a direction with a mirror contributes two entries
  float* scratch;       // [gridDim.y][gridDim.x] (only with loss_out)
  unsigned* counter;    // [1]
  float* loss_out;      // [1] or null (no in-launch combine)
};
__global__ __launch_bounds__(256) void lse_merge_kernel(const MergeBatch batch) {
  const MergeProb& p = batch.p[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float2 quarter[4][64];
  __shared__ unsigned ticket_s;
  const int i = blockIdx.x * 64 + lane;
  const bool active = (int)(blockIdx.x * 64) < p.N;
  float block_sum = 0.f;
  if (active) {
    float mx = -INFINITY, l = 0.f;
    if (i < p.N) {
      // four loads in flight per lane, then the online merge (v_exp_f32: arguments <= 0)
      int t = wave;
      for (; t + 12 < p.n_part; t += 16) {
        float2 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = p.part[(size_t)(t + 4 * q) * p.part_ld + i];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (v[q].x > -INFINITY) {
            const float nm = fmaxf(mx, v[q].x);
            l = l * fast_exp2(mx - nm) + v[q].y * fast_exp2(v[q].x - nm);
            mx = nm;
          }
      }
      for (; t < p.n_part; t += 4) {
        const float2 v = p.part[(size_t)t * p.part_ld + i];
        if (v.x > -INFINITY) {
          const float nm = fmaxf(mx, v.x);
          l = l * fast_exp2(mx - nm) + v.y * fast_exp2(v.x - nm);
          mx = nm;
        }
      }
    }
    quarter[wave][lane] = make_float2(mx, l);
    __syncthreads();
    if (wave == 0) {
      float local = 0.f;
      if (i < p.N) {
        float m = -INFINITY;
#pragma unroll
        for (int q = 0; q < 4; ++q) m = fmaxf(m, quarter[q][lane].x);
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // fixed order
          const float2 v = quarter[q][lane];
          if (v.x > -INFINITY) sum += v.y * fast_exp2(v.x - m);
        }
        const float lse = (m + log2f(sum)) * 0.6931471805599453f;  // partials are in the log2 domain
        p.lse[i] = lse;
        local = lse - p.diag[i];
      }
      block_sum = wave_sum(local);
      if (lane == 0) p.loss_part[blockIdx.x] = block_sum;
    }
  }
  if (batch.loss_out == nullptr) return;
  const unsigned n_blocks = gridDim.x * gridDim.y;
  if (tid == 0) {
    st_agent(batch.scratch + blockIdx.y * gridDim.x + blockIdx.x, active ? p.w * block_sum : 0.f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ticket_s = __hip_atomic_fetch_add(batch.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // A count left behind by an aborted launch (or a workspace bound to a recycled stream handle) shows up as a ticket beyond
  // the grid: fail loudly -- a NaN loss -- instead of returning a stale or never-written value, and leave the counter
  // re-armed once the last workgroup of this launch has drawn.
  if (ticket_s >= n_blocks) {
    if (tid == 0) {
      *batch.loss_out = __builtin_nanf("");
      if ((ticket_s + 1) % n_blocks == 0) __hip_atomic_store(batch.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (ticket_s != n_blocks - 1) return;
  if (tid < 64) {   // last arriver: fixed-order sum over all block sums
    float t = 0.f;
    for (unsigned k = tid; k < n_blocks; k += 64) t += ld_agent(batch.scratch + k);
    t = wave_sum(t);
    if (tid == 0) {
      *batch.loss_out = t;
      __hip_atomic_store(batch.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// alignment: loss_r = pos/npos + neg/nneg per owned row; block sums to loss_part (the caller scales by 1/M)
struct AlignReduceProb {
  const float2* part;
  int part_ld, tiles_m, N, M, row0;
  const int* hmax;
  float* loss_part;
};
struct AlignReduceBatch {
  AlignReduceProb p[MAX_PROBS];
};
__global__ __launch_bounds__(256) void align_reduce_kernel(const AlignReduceBatch batch) {
  const AlignReduceProb& p = batch.p[blockIdx.y];
  if (blockIdx.x * 256 >= p.N) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float local = 0.f;
  if (i < p.N) {
    float pos = 0.f, neg = 0.f;
    for (int t = 0; t < p.tiles_m; ++t) {
      const float2 v = p.part[(size_t)t * p.part_ld + i];
      pos += v.x;
      neg += v.y;
    }
    const int row = p.row0 + i;
    const float npos = (float)(p.hmax[row] - row);
    local = pos / npos + neg / ((float)p.M - npos);
  }
  __shared__ float red[4];
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) p.loss_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[k] = w[k] * sum(ptr[k][0..cnt[k]))   (separate)   or   out[0] = sum_k of that   (combined)
struct CombineArgs {
  const float* ptr[2 * MAX_PROBS];
  int cnt[2 * MAX_PROBS];
  float w[2 * MAX_PROBS];
  int n;
  int separate;
};
__global__ __launch_bounds__(64) void reduce_sums_kernel(const CombineArgs a, float* out) {
  float total = 0.f;
  for (int k = 0; k < a.n; ++k) {
    float t = 0.f;
    for (int i = threadIdx.x; i < a.cnt[k]; i += 64) t += a.ptr[k][i];
    t = wave_sum(t) * a.w[k];
    if (a.separate) {
      if (threadIdx.x == 0) out[k] = t;
    } else {
      total += t;
    }
  }
  if (!a.separate && threadIdx.x == 0) out[0] = total;
}

// ------------------------------------------------------------------ pack / transpose
// one wave per destination row: gather, optional L2 normalise (eps 1e-12, F.normalize), cast, zero pad
template <typename S, typename T>
__global__ __launch_bounds__(256) void pack_rows_kernel(const S* __restrict__ src, int d, const int32_t* __restrict__ idx,
                                                        int r, int normalize, T* __restrict__ dst, int r_pad, int k_pad,
                                                        float* __restrict__ norm_out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= r_pad) return;
  T* out = dst + (size_t)row * k_pad;
  if (row >= r) {
    for (int c = lane * 4; c < k_pad; c += 256) Vec4<T>::store(out + c, make_float4(0.f, 0.f, 0.f, 0.f));
    if (norm_out != nullptr && lane == 0) norm_out[row] = 0.f;
    return;
  }
  const S* in = src + (size_t)(idx ? idx[row] : row) * d;
  const bool vec = (d & 3) == 0;
  float inv = 1.f;
  if (normalize) {
    float ss = 0.f;
    if (vec) {
      for (int c = lane * 4; c < d; c += 256) {
        const float4 v = Vec4<S>::load(in + c);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
    } else {
      for (int c = lane; c < d; c += 64) {
        const float v = to_f32(in[c]);
        ss += v * v;
      }
    }
    ss = wave_sum(ss);
    inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
  }
  float nrm2 = 0.f;
  for (int c = lane * 4; c < k_pad; c += 256) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec) {
      if (c < d) v = Vec4<S>::load(in + c);
    } else {
      if (c + 0 < d) v.x = to_f32(in[c + 0]);
      if (c + 1 < d) v.y = to_f32(in[c + 1]);
      if (c + 2 < d) v.z = to_f32(in[c + 2]);
      if (c + 3 < d) v.w = to_f32(in[c + 3]);
    }
    v.x *= inv;
    v.y *= inv;
    v.z *= inv;
    v.w *= inv;
    Vec4<T>::store(out + c, v);
    // the norm of what the MFMA will multiply: the values as rounded to T
    const float q0 = to_f32(from_f32<T>(v.x)), q1 = to_f32(from_f32<T>(v.y)), q2 = to_f32(from_f32<T>(v.z)), q3 = to_f32(from_f32<T>(v.w));
    nrm2 += q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
  }
  if (norm_out != nullptr) {
    nrm2 = wave_sum(nrm2);
    if (lane == 0) norm_out[row] = sqrtf(nrm2);
  }
}

// Batched pack: gather + optional L2 normalise + cast + zero pad AND the transposed copy, for several operands in ONE
// launch (grid.y = operand).  A workgroup owns 64 destination rows: row norms first (one wave per 16 rows), then the
// row block walks the k axis in 64-column steps through a [64][64] LDS tile that is written out both ways.
struct PackEntry {
  const void* src;
  const int32_t* idx;
  void* dst;
  void* dstT;   // or null
  int r, r_pad, normalize, ldt;
  float* norm;  // [r_pad] L2 norms of the packed rows, or null
};
struct PackBatch {
  PackEntry e[MAX_PROBS];
};
template <typename S, typename T>
__global__ __launch_bounds__(256) void pack_tr_kernel(const PackBatch batch, int d, int k_pad) {
  // 16 destination rows per workgroup (many small workgroups: the operands are a few MB).  Thread (row = tid >> 4,
  // c4 = tid & 15) owns 4 consecutive columns of every 64-column step.
  const PackEntry& en = batch.e[blockIdx.y];
  const int r0 = blockIdx.x * 16;
  if (r0 >= en.r_pad) return;
  __shared__ float inv_s[16];
  __shared__ T tile[16][68];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const S* src = static_cast<const S*>(en.src);
  const bool vec = (d & 3) == 0;
  if (en.normalize) {
    for (int rr = wave * 4; rr < wave * 4 + 4; ++rr) {
      const int row = r0 + rr;
      float ss = 0.f;
      if (row < en.r) {
        const S* in = src + (size_t)(en.idx ? en.idx[row] : row) * d;
        if (vec) {
          for (int c = lane * 4; c < d; c += 256) {
            const float4 v = Vec4<S>::load(in + c);
            ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
          }
        } else {
          for (int c = lane; c < d; c += 64) {
            const float v = to_f32(in[c]);
            ss += v * v;
          }
        }
      }
      ss = wave_sum(ss);
      if (lane == 0) inv_s[rr] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
    }
    __syncthreads();
  }
  const int rr = tid >> 4, c4 = (tid & 15) * 4;
  const int row = r0 + rr;
  const S* in = row < en.r ? src + (size_t)(en.idx ? en.idx[row] : row) * d : nullptr;
  const float inv = en.normalize ? inv_s[rr] : 1.f;
  T* dst = static_cast<T*>(en.dst) + (size_t)row * k_pad;
  T* dstT = static_cast<T*>(en.dstT);
  const int kr = tid >> 2, s4 = (tid & 3) * 4;   // transposed write: k row kr of the step, tile rows s4 .. s4 + 3
  float nrm2 = 0.f;
  for (int k0 = 0; k0 < k_pad; k0 += 64) {
    const int c = k0 + c4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in != nullptr) {
      if (vec) {
        if (c < d) v = Vec4<S>::load(in + c);
      } else {
        if (c + 0 < d) v.x = to_f32(in[c + 0]);
        if (c + 1 < d) v.y = to_f32(in[c + 1]);
        if (c + 2 < d) v.z = to_f32(in[c + 2]);
        if (c + 3 < d) v.w = to_f32(in[c + 3]);
      }
    }
    v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
    Vec4<T>::store(dst + c, v);
    {
      const float q0 = to_f32(from_f32<T>(v.x)), q1 = to_f32(from_f32<T>(v.y)), q2 = to_f32(from_f32<T>(v.z)), q3 = to_f32(from_f32<T>(v.w));
      nrm2 += q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
    }
    if (dstT != nullptr) {
      Vec4<T>::store(&tile[rr][c4], v);
      __syncthreads();
      Vec4<T>::store(dstT + (size_t)(k0 + kr) * en.ldt + r0 + s4,
                     make_float4(to_f32(tile[s4][kr]), to_f32(tile[s4 + 1][kr]), to_f32(tile[s4 + 2][kr]), to_f32(tile[s4 + 3][kr])));
      __syncthreads();
    }
  }
  if (en.norm != nullptr) {   // the 16 threads of a row are 16 consecutive lanes
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) nrm2 += __shfl_xor(nrm2, o);
    if ((tid & 15) == 0) en.norm[row] = sqrtf(nrm2);
  }
}

// dstT[k][p] = dst[p][k]; 64x64 tiles through LDS (+1 padding), both dims multiples of 64
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int rows, int cols,
                                                        int ld_out) {
  __shared__ T tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int y = ty; y < 64; y += 4) tile[y][tx] = in[(size_t)(r0 + y) * cols + c0 + tx];
  __syncthreads();
  for (int y = ty; y < 64; y += 4) out[(size_t)(c0 + y) * ld_out + r0 + tx] = tile[tx][y];
}

// ------------------------------------------------------------------ finalize
// one wave per owned row: dy = coef * sum_splits slab ; optional F.normalize backward ; scatter to user grad
template <typename U>
__global__ __launch_bounds__(256) void grad_finalize_kernel(const FinBatch batch, const float* __restrict__ scale_ptr,
                                                            const float* __restrict__ upstream, const DsBatch ds, float* ds_out, int n_dirs) {
  extern __shared__ __attribute__((aligned(16))) float rowbuf[];  // [4][slab_ld]
  if ((int)blockIdx.y == n_dirs) {
    // extra grid row: d loss / d scale = upstream * sum_k kappa_k * sum(tile partials of the gradient-tile pass)
    if (blockIdx.x != 0) return;
    float local = 0.f;
    for (int k = 0; k < ds.n_probs; ++k) {
      float t = 0.f;
      for (int i = threadIdx.x; i < ds.n[k]; i += 256) t += ds.part[k][i];
      local += ds.kappa[k] * t;
    }
    local = wave_sum(local);
    if ((threadIdx.x & 63) == 0) rowbuf[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) *ds_out += (*upstream) * (rowbuf[0] + rowbuf[1] + rowbuf[2] + rowbuf[3]);
    return;
  }
  const FinProb& p = batch.p[blockIdx.y];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= p.r) return;
  const int d = batch.d;
  const float coef = p.kappa * (*scale_ptr) * (*upstream);
  float* buf = rowbuf + (size_t)wave * p.slab_ld;
  const int dst_row = p.dx_rows ? p.dx_rows[i] : i;
  const U* x = reinterpret_cast<const U*>(p.src) + (size_t)dst_row * d;
  const bool vec = (d & 3) == 0;  // 4-element (16/8-byte) lanes; slab rows are k_pad wide, always 16-byte aligned
  float dot = 0.f, ss = 0.f;
  for (int c = lane * 4; c < d; c += 256) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < p.n_split; ++s) {
      const float4 t = *reinterpret_cast<const float4*>(p.slab + (size_t)s * p.split_stride + (size_t)i * p.slab_ld + c);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef;
    for (int k = 0; k < p.n_extra; ++k) {
      const float ck = p.kappa_extra[k] * (*scale_ptr) * (*upstream);
      const float4 t = *reinterpret_cast<const float4*>(p.extra[k] + (size_t)i * p.slab_ld + c);
      v.x = fmaf(ck, t.x, v.x); v.y = fmaf(ck, t.y, v.y); v.z = fmaf(ck, t.z, v.z); v.w = fmaf(ck, t.w, v.w);
    }
    *reinterpret_cast<float4*>(buf + c) = v;
    if (p.normalize) {
      float xv[4] = {0.f, 0.f, 0.f, 0.f};
      if (vec) {
        const float4 t = Vec4<U>::load(x + c);
        xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
      } else {
        for (int e = 0; e < 4; ++e)
          if (c + e < d) xv[e] = to_f32(x[c + e]);
      }
      const float dv[4] = {v.x, v.y, v.z, v.w};
      for (int e = 0; e < 4; ++e)
        if (c + e < d) {
          dot += dv[e] * xv[e];
          ss += xv[e] * xv[e];
        }
    }
  }
  float inv = 1.f, proj = 0.f;
  if (p.normalize) {
    dot = wave_sum(dot);
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    inv = 1.f / fmaxf(nrm, 1e-12f);
    // y = x*inv ; dx = (dy - y (y.dy)) * inv   (norm clamped at eps => constant, no projection)
    proj = (nrm > 1e-12f) ? dot * inv * inv : 0.f;
  }
  for (int c = lane * 4; c < d; c += 256) {
    float4 v = *reinterpret_cast<const float4*>(buf + c);
    float o[4] = {v.x, v.y, v.z, v.w};
    if (p.normalize) {
      for (int e = 0; e < 4; ++e)
        if (c + e < d) o[e] = (o[e] - to_f32(x[c + e]) * proj) * inv;
    }
    if (p.accumulate && p.exclusive && vec) {   // one writer per element in this launch, earlier launches ordered by the stream
      float4* q = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.dx) + (size_t)dst_row * d + c);
      float4 t = *q;
      t.x += o[0]; t.y += o[1]; t.z += o[2]; t.w += o[3];
      *q = t;
    } else if (p.accumulate) {
      for (int e = 0; e < 4; ++e)
        if (c + e < d) atomicAdd(reinterpret_cast<float*>(p.dx) + (size_t)dst_row * d + c + e, o[e]);
    } else if (vec) {
      Vec4<U>::store(reinterpret_cast<U*>(p.dx) + (size_t)dst_row * d + c, make_float4(o[0], o[1], o[2], o[3]));
    } else {
      for (int e = 0; e < 4; ++e)
        if (c + e < d) reinterpret_cast<U*>(p.dx)[(size_t)dst_row * d + c + e] = from_f32<U>(o[e]);
    }
  }
}

// ------------------------------------------------------------------ host side
struct Plan {
  int bm, bn;      // tile of the similarity kernels: BM columns (P rows) x BN owned rows (Q rows)
  int bm_g, bn_g;  // tile of the gradient GEMM: BM_g of the k_pad output columns x BN_g owned rows
  int n_split;     // split-K factor of the gradient GEMM
};
// 128x128 when that gives >= `want` tiles, else 64x64 (small problems are latency-bound: more, smaller blocks win;
// 128x64 is kept as an experiment override).
static void pick_tile(long rows_p, long rows_q, int n_probs, long want, int* bm, int* bn) {
  if (const char* e = MMK_DBG_ENV("MMK_TILE")) {  // experiment override: 64 | 12864 | 128
    const int v = atoi(e);
    *bm = v == 64 ? 64 : 128;
    *bn = v == 128 ? 128 : 64;
    return;
  }
  const long t128 = cdiv((int)rows_p, 128) * (long)cdiv((int)rows_q, 128) * n_probs;
  const long t12864 = cdiv((int)rows_p, 128) * (long)cdiv((int)rows_q, 64) * n_probs;
  (void)t12864;  // measured at N = 1024 (rocprofv3): 64x64 9.2-10.2 us < 128x64 10.2-11.7 us < 128x128 12.7-16.6 us
  if (t128 >= want) { *bm = 128; *bn = 128; }
  else { *bm = 64; *bn = 64; }
}
static Plan make_plan(int r_max, int c_max, int k_pad, int n_dirs, int compute) {
  Plan pl;
  pick_tile(c_max, r_max, n_dirs, 256, &pl.bm, &pl.bn);
  const int bk = compute == MMK_COMPUTE_BF16 ? 64 : 32;
  const int c_pad = round_up(c_max, 128);
  pick_tile(k_pad, r_max, n_dirs, 128, &pl.bm_g, &pl.bn_g);
  const long gt = (long)cdiv(k_pad, pl.bm_g) * cdiv(r_max, pl.bn_g) * n_dirs;
  int split = (int)((768 + gt - 1) / gt);
  const int max_split = c_pad / (2 * bk) > 0 ? c_pad / (2 * bk) : 1;  // >= 2 k-steps per split
  if (split > max_split) split = max_split;
  if (split > 16) split = 16;
  if (split < 1) split = 1;
  pl.n_split = split;
  // Row-sharded shapes (few G row tiles, a long contraction: R = 1024 x C = 8192 on each of 8 ranks).  With 128 x 128 tiles and a
  // split count that makes (directions x splits) a multiple of 8, every (direction, split) unit sits on ONE XCD (unit_map in
  // gemm_nt_kernel) and its K range of G and of Y^T is fetched from HBM once; the tile-range map has all eight XCDs read every
  // Y^T range.  Taken when it fills one round of the chip (448 .. 640 workgroups); the slabs it adds (splits x R x D f32) are
  // less than the Y^T re-reads it removes from 4096 columns on (measured, profiles/r05_pmc_traffic_shard.json: the rank share's HBM-side
  // bytes 346 -> 272 MB at C = 8192, 202 -> 185 MB at 4096, but 130 -> 142 MB at 2048).
  if (c_pad >= 4096 && !MMK_DBG_ENV("MMK_TILE") && !(MMK_DBG_ENV("MMK_GRAD_UNIT_MAP") && atoi(MMK_DBG_ENV("MMK_GRAD_UNIT_MAP")) == 0)) {
    const long gt128 = (long)cdiv(k_pad, 128) * cdiv(r_max, 128) * n_dirs;
    for (int s8 = 4; s8 <= 16 && s8 <= max_split; s8 += 4)
      if ((n_dirs * s8) % 8 == 0 && gt128 * s8 >= 448 && gt128 * s8 <= 640) {
        pl.bm_g = pl.bn_g = 128;
        pl.n_split = s8;
        break;
      }
  }
  return pl;
}

// loader / pipeline depth selection (MMK_LOADER=reg|dma, MMK_STAGES=2|3|4 override the defaults; used for A/B runs)
struct LoaderCfg {
  int loader, stages;
};
static LoaderCfg loader_cfg() {
  static LoaderCfg cfg = [] {
    LoaderCfg c{LOADER_DMA, 2};
    if (const char* e = MMK_DBG_ENV("MMK_LOADER")) c.loader = (e[0] == 'r') ? LOADER_REG : LOADER_DMA;
    if (const char* e = MMK_DBG_ENV("MMK_STAGES")) c.stages = atoi(e);
    if (c.loader == LOADER_REG) c.stages = 2;
    if (c.stages < 2 || c.stages > 4) c.stages = 2;
    return c;
  }();
  return cfg;
}

template <typename T, int BM, int BN, int EPI, int LOADER, int NSTAGE, int WM = 2>
static int launch_one(const ProbBatch& b, dim3 grid, const float* scale, hipStream_t st) {
  constexpr int bytes = NSTAGE * (BM + BN) * 128;
  auto kern = gemm_nt_kernel<T, BM, BN, EPI, LOADER, NSTAGE, WM>;
  static bool attr_set = false;
  if (bytes > 64 * 1024 && !attr_set) {
    MMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    attr_set = true;
  }
  {
    constexpr int kid = (EPI == EPI_STATS || EPI == EPI_ALIGN_STATS) ? MMK_K_SIM_STATS
                        : (EPI == EPI_GRAD || EPI == EPI_ALIGN_GRAD) ? MMK_K_SIM_GRAD : MMK_K_GRAD_GEMM;
    ProfEvents pe(kid);  // null events unless profiling: then the dispatch itself is time-stamped
    hipExtLaunchKernelGGL(kern, grid, dim3(128 * WM), bytes, st, pe.start, pe.stop, 0, b, scale);
  }
  MMK_LAUNCH_CHECK();
  return 0;
}

template <typename T, int BM, int BN, int EPI>
static int launch_tile(const ProbBatch& b, dim3 grid, const float* scale, hipStream_t st) {
  LoaderCfg c = loader_cfg();
  if (c.loader == LOADER_REG) return launch_one<T, BM, BN, EPI, LOADER_REG, 2>(b, grid, scale, st);
  // 2 LDS stages by default: 128^2 tiles then fit 2 blocks/CU (3-4 stages measured 1.4-1.7x slower at N = 8192);
  // for the latency-bound 64^2 tiles 4 stages were within noise of 2 (N = 512..2048), so keep the smaller footprint.
  if (c.stages == 2) return launch_one<T, BM, BN, EPI, LOADER_DMA, 2>(b, grid, scale, st);
  if (c.stages == 3) return launch_one<T, BM, BN, EPI, LOADER_DMA, 3>(b, grid, scale, st);
  return launch_one<T, BM, BN, EPI, LOADER_DMA, 4>(b, grid, scale, st);
}

template <typename T, int EPI>
static int launch_gemm(const ProbBatch& b, int n_probs, int bm, int bn, int max_tiles, const float* scale, hipStream_t st) {
  dim3 grid(max_tiles, 1, n_probs * (EPI == EPI_PLAIN ? b.n_split : 1));
  if (bm == 256 && bn == 128) {   // the eight-wave tile of the statistics pass (see stats_tile)
    if (EPI == EPI_STATS) return launch_one<T, 256, 128, EPI_STATS, LOADER_DMA, 3, 4>(b, grid, scale, st);
    MMK_REQUIRE(false, "the 256 x 128 tile is built for the statistics pass only");
  }
  if (bm == 128 && bn == 128) return launch_tile<T, 128, 128, EPI>(b, grid, scale, st);
  if (bm == 128 && bn == 64) return launch_tile<T, 128, 64, EPI>(b, grid, scale, st);
  return launch_tile<T, 64, 64, EPI>(b, grid, scale, st);
}

// ---- forward statistics of row-sharded directions as ONE streaming launch (csrc/clip_bwd.hip, clip_fwd_shard_kernel)
// A direction takes it when it is a row shard with no mirrored partner (W > 1: column statistics come through the all-reduce): bf16,
// k_pad = 512, cross-entropy mode, >= 1024 columns, and the call's directions of that kind make half a chip of workgroups.  The
// column split minimises (rounds of 256 workgroups) x (fixed cost of ~4 tile times + tiles); one partial per (split, row) lands in
// the direction's `part` rows 0 .. n_split - 1 (the caller sized it for ceil(c / 64) column tiles).
struct FwdShardPlan {
  bool shard[MAX_PROBS];
  int n_split;
};
static FwdShardPlan fwd_shard_plan(const mmk_clip_dir* dirs, int n_dirs, int k_pad, int compute) {
  FwdShardPlan fp{};
  if (MMK_DBG_ENV("MMK_CLIP_FWD_SHARD") && atoi(MMK_DBG_ENV("MMK_CLIP_FWD_SHARD")) == 0) return fp;   // A/B in debug-switch builds
  if (compute != MMK_COMPUTE_BF16 || k_pad != 512) return fp;
  int row_blocks = 0, c_min = 1 << 30;
  for (int k = 0; k < n_dirs; ++k) {
    const mmk_clip_dir& d = dirs[k];
    fp.shard[k] = d.mode == 0 && d.mirror_part == nullptr && d.c >= 1024;
    if (fp.shard[k]) {
      row_blocks += cdiv(d.r, 64);
      c_min = std::min(c_min, d.c);
    }
  }
  if (row_blocks == 0) return fp;
  const int ns_max = std::max(1, std::min(16, cdiv(c_min, 64) / 2));
  int ns = 1;
  long best = -1;
  for (int s = 1; s <= ns_max; ++s) {
    const long cost = (long)cdiv(row_blocks * s, 256) * (4 + cdiv(cdiv(c_min, 64), s));
    if (best < 0 || cost < best) {
      best = cost;
      ns = s;
    }
  }
  if ((long)row_blocks * ns < 128) {
    for (int k = 0; k < n_dirs; ++k) fp.shard[k] = false;
    return fp;
  }
  fp.n_split = ns;
  return fp;
}

template <typename T>
static int clip_forward_impl(const mmk_clip_dir* dirs, int n_dirs, int k_pad, const float* scale, const float* loss_w, float* loss_out,
                             int32_t* tickets, int n_tickets, hipStream_t st) {
  int r_max = 0, c_max = 0;
  for (int k = 0; k < n_dirs; ++k) {
    r_max = std::max(r_max, dirs[k].r);
    c_max = std::max(c_max, dirs[k].c);
  }
  Plan pl = make_plan(r_max, c_max, k_pad, n_dirs, sizeof(T) == 2 ? MMK_COMPUTE_BF16 : MMK_COMPUTE_F32);
  // Experiment (MMK_STATS_TILE=256, off by default): 256 x 128 tiles on eight waves, three stages, one workgroup per CU -- 3/4 of
  // the bytes into LDS per output, 96 KiB of them in flight per CU instead of 64.  Measured at N = 8192: loads alone 62 us (as
  // with two 128 x 128 workgroups per CU), main loop 106 us instead of 77, whole kernel 121 instead of 109: one workgroup's
  // eight waves in barrier lock-step hide the MFMAs worse than two independent four-wave workgroups do.
  if (dirs[0].mode == 0 && loader_cfg().loader == LOADER_DMA && MMK_DBG_ENV("MMK_TILE") == nullptr &&
      MMK_DBG_ENV("MMK_STATS_TILE") && atoi(MMK_DBG_ENV("MMK_STATS_TILE")) == 256 &&
      (long)cdiv(c_max, 256) * cdiv(r_max, 128) * n_dirs >= 256) {
    pl.bm = 256;
    pl.bn = 128;
  }
  ProbBatch b;
  b.unit_map = 0;
  MergeBatch mb;
  AlignReduceBatch ab;
  int max_tiles = 0, n_red = 0, r_red_max = 0, n_tile_dirs = 0;
  const bool align = dirs[0].mode == 1;
  const FwdShardPlan sp = fwd_shard_plan(dirs, n_dirs, k_pad, sizeof(T) == 2 ? MMK_COMPUTE_BF16 : MMK_COMPUTE_F32);
  FwdShardBatch fs;
  fs.n_probs = 0;
  fs.n_split = std::max(sp.n_split, 1);
  fs.cols_per_split = round_up(cdiv(round_up(c_max, 64), fs.n_split), 64);
  fs.row_blocks = 0;
  fs.groups = 1;
  fs.rows_per_group = 0;
  for (int k = 0; k < n_dirs; ++k) {
    const mmk_clip_dir& d = dirs[k];
    if (sp.shard[k]) {
      FwdShardProb& q = fs.p[fs.n_probs++];
      q.x = static_cast<const bf16_t*>(d.x); q.y = static_cast<const bf16_t*>(d.y);
      q.part = reinterpret_cast<float2*>(d.part); q.diag = d.diag;
      q.r = d.r; q.c = d.c; q.part_ld = d.r; q.label_off = d.label_off;
      fs.row_blocks = std::max(fs.row_blocks, cdiv(d.r, 64));
      mb.p[n_red] = MergeProb{q.part, q.part_ld, fs.n_split, d.r, d.diag, d.lse, d.loss_part, loss_w ? loss_w[n_red] : 0.f};
      ++n_red;
      r_red_max = std::max(r_red_max, d.r);
      continue;
    }
    Prob& p = b.p[n_tile_dirs];
    p = Prob{};
    p.P = static_cast<const char*>(d.y);
    p.Q = static_cast<const char*>(d.x);
    p.M = d.c;
    p.N = d.r;
    p.K = k_pad;
    p.ldp = p.ldq = k_pad;
    p.tiles_m = cdiv(d.c, pl.bm);
    p.tiles_n = cdiv(d.r, pl.bn);
    p.part = reinterpret_cast<float2*>(d.part);
    p.part_ld = d.r;
    p.diag = d.diag;
    p.label_off = d.label_off;
    p.hmax = d.hmax;
    if (d.x_norm != nullptr && d.y_norm != nullptr) {
      p.qn = d.x_norm;
      p.pn = d.y_norm;
    }
    max_tiles = std::max(max_tiles, p.tiles_m * p.tiles_n);
    ab.p[n_tile_dirs] = AlignReduceProb{p.part, p.part_ld, p.tiles_m, d.r, d.c, d.label_off, d.hmax, d.loss_part};
    ++n_tile_dirs;
    if (!align) {
      mb.p[n_red] = MergeProb{p.part, p.part_ld, p.tiles_m, d.r, d.diag, d.lse, d.loss_part, loss_w ? loss_w[n_red] : 0.f};
      ++n_red;
      r_red_max = std::max(r_red_max, d.r);
      if (d.mirror_part != nullptr) {
        // the mirrored direction's rows are this direction's columns: partials per (row tile, wn half), positive = same diagonal
        p.mpart = reinterpret_cast<float2*>(d.mirror_part);
        p.mpart_ld = d.c;
        mb.p[n_red] = MergeProb{p.mpart, p.mpart_ld, 2 * p.tiles_n, d.c, d.diag, d.mirror_lse, d.mirror_loss_part, loss_w ? loss_w[n_red] : 0.f};
        ++n_red;
        r_red_max = std::max(r_red_max, d.c);
      }
    }
  }
  b.n_split = 1;
  b.n_probs = n_tile_dirs;
  b.dbg = MMK_DBG_ENV("MMK_SIM_DBG") ? atoi(MMK_DBG_ENV("MMK_SIM_DBG")) : 0;
  if (fs.n_probs > 0) {
    if (int rc = launch_clip_fwd_shard(fs, scale, st)) return rc;
  }
  if (n_tile_dirs > 0) {
    int rc = align ? launch_gemm<T, EPI_ALIGN_STATS>(b, n_tile_dirs, pl.bm, pl.bn, max_tiles, scale, st)
                   : launch_gemm<T, EPI_STATS>(b, n_tile_dirs, pl.bm, pl.bn, max_tiles, scale, st);
    if (rc) return rc;
  }
  {
    if (align) {
      ProfScope ps(MMK_K_LSE_REDUCE, st);
      hipLaunchKernelGGL(align_reduce_kernel, dim3(cdiv(r_max, 256), n_dirs), dim3(256), 0, st, ab);
    } else {
      const dim3 grid(cdiv(r_red_max, 64), n_red);
      mb.loss_out = nullptr;
      mb.counter = nullptr;
      mb.scratch = nullptr;
      if (loss_out != nullptr) {
        MMK_REQUIRE(tickets != nullptr && (long)grid.x * grid.y + 1 <= n_tickets, "ticket workspace too small (see mmk_clip_tickets)");
        mb.loss_out = loss_out;
        mb.counter = reinterpret_cast<unsigned*>(tickets);
        mb.scratch = reinterpret_cast<float*>(tickets) + 1;
      }
      ProfEvents pe(MMK_K_LSE_REDUCE);   // dispatch-stamped (see launch_grad_finalize)
      hipExtLaunchKernelGGL(lse_merge_kernel, grid, dim3(256), 0, st, pe.start, pe.stop, 0, mb);
    }
    MMK_LAUNCH_CHECK();
  }
  return 0;
}

// ---- the one-kernel recompute-G backward of row-sharded directions (csrc/clip_bwd.hip)
// A direction takes it when it is a SHARD with a tile pass of its own: bf16, k_pad = 512, no alignment term, not one half of a mirrored
// pair (those share one tile pass and hand G^T over) and at least 1024 columns -- and when the call's directions of that kind, with
// the column split chosen for them, make at least half a chip of (row block, split) workgroups.  The split is this kernel's own:
// about 256 workgroups (one per CU: its LDS admits no second), within the slab count the caller sized by mmk_clip_plan at the
// call's largest shape, at least two 64-column tiles each, whole rounds of the chip preferred.  Reads shapes, modes and pairing only (mmk_clip_backward_plan answers
// before the buffers exist).
struct BwdFusedPlan {
  bool fused[MAX_PROBS];
  int n_split;   // of the fused directions (0: none)
};
static BwdFusedPlan bwd_fused_plan(const mmk_clip_dir* dirs, int n_dirs, int k_pad, int compute) {
  BwdFusedPlan fp{};
  if (MMK_DBG_ENV("MMK_CLIP_BWD_FUSED") && atoi(MMK_DBG_ENV("MMK_CLIP_BWD_FUSED")) == 0) return fp;   // A/B in debug-switch builds
  if (compute != MMK_COMPUTE_BF16 || k_pad != 512) return fp;
  int r_max = 0, c_max = 0, row_blocks = 0, c_pad_min = 1 << 30;
  for (int k = 0; k < n_dirs; ++k) {
    r_max = std::max(r_max, dirs[k].r);
    c_max = std::max(c_max, dirs[k].c);
  }
  for (int k = 0; k < n_dirs; ++k) {
    const mmk_clip_dir& d = dirs[k];
    bool ok = d.mode == 0 && !d.g_ready && !d.g_transposed && d.gT == nullptr && d.c >= 1024;
    for (int j = 0; j < n_dirs && ok; ++j)   // a mirrored partner reads this direction's G (transposed, csrc/wgrad.hip): G must exist then
      if (j != k && dirs[j].g_ready && d.g != nullptr && dirs[j].g == d.g) ok = false;
    fp.fused[k] = ok;
    if (ok) {
      row_blocks += cdiv(d.r, 64);
      c_pad_min = std::min(c_pad_min, round_up(d.c, 128));
    }
  }
  if (row_blocks == 0) return fp;
  int32_t cap = 1;
  mmk_clip_plan(r_max, c_max, k_pad, compute, nullptr, nullptr, &cap);
  // the split that minimises (rounds of 256 workgroups) x (per-workgroup fixed cost + its tiles); the fixed cost -- prologue, slab
  // stores -- is about eight tile times (measured: ~17 us against ~2 us per 64-column tile)
  const int ns_max = std::max(1, std::min((int)cap, c_pad_min / 128));
  int ns = 1;
  long best = -1;
  for (int s = 1; s <= ns_max; ++s) {
    const long cost = (long)cdiv(row_blocks * s, 256) * (8 + cdiv(c_pad_min / 64, s));
    if (best < 0 || cost < best) {
      best = cost;
      ns = s;
    }
  }
  if ((long)row_blocks * ns < 128) {   // too little work for this form: the two-launch path tiles finer
    for (int k = 0; k < n_dirs; ++k) fp.fused[k] = false;
    return fp;
  }
  fp.n_split = ns;
  return fp;
}

template <typename T, typename U>
static int clip_backward_impl(const mmk_clip_dir* dirs, int n_dirs, int k_pad, int d_user, const float* scale,
                              const float* upstream, float* dscale_out, hipStream_t st) {
  int r_max = 0, c_max = 0;
  for (int k = 0; k < n_dirs; ++k) {
    r_max = std::max(r_max, dirs[k].r);
    c_max = std::max(c_max, dirs[k].c);
  }
  const Plan pl = make_plan(r_max, c_max, k_pad, n_dirs, sizeof(T) == 2 ? MMK_COMPUTE_BF16 : MMK_COMPUTE_F32);
  const int bk = Atom<T>::BK;
  ProbBatch gb, xb;
  FinBatch fb;
  DsBatch db;
  int max_tiles_g = 0, max_tiles_x = 0, max_r = 0, n_tile_probs = 0, n_x = 0, n_tn = 0;
  int tn[MAX_PROBS];
  db.n_probs = 0;
  const int c_pad_max = round_up(c_max, 128);
  const int k_per_split = round_up(cdiv(c_pad_max, pl.n_split), bk);
  const BwdFusedPlan fp = bwd_fused_plan(dirs, n_dirs, k_pad, sizeof(T) == 2 ? MMK_COMPUTE_BF16 : MMK_COMPUTE_F32);
  BwdFusedBatch fz;
  fz.n_probs = 0;
  fz.n_split = std::max(fp.n_split, 1);
  fz.cols_per_split = round_up(cdiv(c_pad_max, fz.n_split), 64);
  fz.row_blocks = 0;
  fz.dbg = 0;
  fz.groups = 1;
  fz.rows_per_group = 0;
  for (int k = 0; k < n_dirs; ++k) {
    const mmk_clip_dir& d = dirs[k];
    const int r_pad = round_up(d.r, 128), c_pad = round_up(d.c, 128);
    MMK_REQUIRE(d.ldg >= c_pad && d.ldt >= c_pad, "ldg/ldt must be >= round_up(c, 128)");
    if (fp.fused[k]) {
      MMK_REQUIRE(d.lse_col != nullptr || (d.c_col == 0.f && d.s_col == 0.f), "lse_col required when c_col/s_col != 0");
      BwdFusedProb& q = fz.p[fz.n_probs++];
      q.x = static_cast<const bf16_t*>(d.x); q.y = static_cast<const bf16_t*>(d.y);
      q.lse_row = d.lse; q.lse_col = d.lse_col;
      q.slab = d.slab; q.ds_part = d.ds_part;
      q.r = d.r; q.c = d.c; q.r_pad = r_pad; q.label_off = d.label_off;
      q.c_row = d.c_row; q.c_col = d.c_col; q.c_diag = d.c_diag;
      q.s_row = d.s_row; q.s_col = d.s_col; q.s_diag = d.s_diag;
      fz.row_blocks = std::max(fz.row_blocks, cdiv(d.r, 64));
      db.part[db.n_probs] = d.ds_part;
      db.n[db.n_probs] = cdiv(d.r, 64) * fz.n_split;
      db.kappa[db.n_probs] = d.ds_kappa;
      ++db.n_probs;
      fb.p[k] = FinProb{d.slab, (long)r_pad * k_pad, k_pad, d.r, d.kappa, d.dx, d.dx_rows, d.dx_accumulate, d.src, d.normalize, fz.n_split};
      max_r = std::max(max_r, d.r);
      continue;
    }
    // recompute + G (a direction whose G was written as another direction's G^T has no tile pass of its own)
    Prob p{};
    p.P = static_cast<const char*>(d.y);
    p.Q = static_cast<const char*>(d.x);
    p.M = d.c;
    p.N = d.r;
    p.K = k_pad;
    p.ldp = p.ldq = k_pad;
    p.tiles_m = c_pad / pl.bm;   // cover the zero padding of G up to ldg
    p.tiles_n = d.gT != nullptr ? r_pad / pl.bn : cdiv(d.r, pl.bn);   // ... and of G^T up to ldgt
    p.label_off = d.label_off;
    p.lse_row = d.lse;
    p.lse_col = d.lse_col;
    p.G = static_cast<char*>(d.g);
    p.ldg = d.ldg;
    p.GT = static_cast<char*>(d.gT);
    p.ldgt = d.ldgt;
    p.c_row = d.c_row; p.c_col = d.c_col; p.c_diag = d.c_diag;
    p.s_row = d.s_row; p.s_col = d.s_col; p.s_diag = d.s_diag;
    p.ds_part = d.ds_part;
    p.hmax = d.hmax;
    if (d.x_norm != nullptr && d.y_norm != nullptr && d.mode == 0) {
      p.qn = d.x_norm;
      p.pn = d.y_norm;
    }
    MMK_REQUIRE(d.mode == 1 || d.lse_col != nullptr || (d.c_col == 0.f && d.s_col == 0.f), "lse_col required when c_col/s_col != 0");
    MMK_REQUIRE(d.gT == nullptr || (d.mode == 0 && d.ldgt >= r_pad), "gT needs mode 0 and ldgt >= round_up(r, 128)");
    if (!d.g_ready) {
      gb.p[n_tile_probs++] = p;
      max_tiles_g = std::max(max_tiles_g, p.tiles_m * p.tiles_n);
      db.part[db.n_probs] = d.ds_part;
      db.n[db.n_probs] = p.tiles_m * p.tiles_n;
      db.kappa[db.n_probs] = d.ds_kappa;
      ++db.n_probs;
    }
    if (d.g_transposed) {
      // dX = G_src^T Y with G_src = d.g [c rows, ldg] the OTHER direction's gradient tile matrix: contraction over the rows of
      // both operands = the weight-gradient kernel's form (csrc/wgrad.hip, transposed LDS reads): G^T is never stored
      MMK_REQUIRE(sizeof(T) == 2 && d.mode == 0 && d.g_ready && d.tn_ws != nullptr && d.ldg >= r_pad, "transposed-G direction needs bf16 compute, g_ready, tn_ws and ldg >= round_up(r, 128)");
      tn[n_tn++] = k;
      max_r = std::max(max_r, d.r);
      continue;
    }
    // dX^T[dcol][i] = sum_j yT[dcol][j] * G[i][j]
    Prob& x = xb.p[n_x++];
    x = Prob{};
    x.P = static_cast<const char*>(d.yT);
    x.Q = static_cast<const char*>(d.g);
    x.M = k_pad;
    x.N = d.r;
    x.K = c_pad;
    x.ldp = d.ldt;
    x.ldq = d.ldg;
    x.tiles_m = cdiv(k_pad, pl.bm_g);
    x.tiles_n = cdiv(d.r, pl.bn_g);
    x.slab = d.slab;
    x.slab_ld = k_pad;
    x.slab_split_stride = (long)r_pad * k_pad;
    x.k_per_split = k_per_split;
    max_tiles_x = std::max(max_tiles_x, x.tiles_m * x.tiles_n);
    fb.p[k] = FinProb{d.slab, x.slab_split_stride, k_pad, d.r, d.kappa, d.dx, d.dx_rows, d.dx_accumulate, d.src, d.normalize, pl.n_split};
    max_r = std::max(max_r, d.r);
  }
  gb.n_split = 1;
  gb.unit_map = 0;
  gb.n_probs = n_tile_probs;
  xb.n_probs = n_x;
  gb.dbg = MMK_DBG_ENV("MMK_SIM_DBG") ? (atoi(MMK_DBG_ENV("MMK_SIM_DBG")) & 8) : 0;   // 8 = never take the bounded fast path
  xb.dbg = 0;
  xb.n_split = pl.n_split;
  xb.unit_map = (n_x * pl.n_split) % 8 == 0 && !(MMK_DBG_ENV("MMK_GRAD_UNIT_MAP") && atoi(MMK_DBG_ENV("MMK_GRAD_UNIT_MAP")) == 0);
  fb.d = d_user;
  if (fz.n_probs > 0) {
    int rc = launch_clip_bwd_fused(fz, scale, st);
    if (rc) return rc;
  }
  if (n_tile_probs > 0) {
    int rc = dirs[0].mode == 1 ? launch_gemm<T, EPI_ALIGN_GRAD>(gb, n_tile_probs, pl.bm, pl.bn, max_tiles_g, scale, st)
                               : launch_gemm<T, EPI_GRAD>(gb, n_tile_probs, pl.bm, pl.bn, max_tiles_g, scale, st);
    if (rc) return rc;
  }
  if (n_x > 0) {
    int rc = launch_gemm<T, EPI_PLAIN>(xb, n_x, pl.bm_g, pl.bn_g, max_tiles_x, scale, st);
    if (rc) return rc;
  }
  int ld_max = k_pad;
  for (int t = 0; t < n_tn; ++t) {
    const mmk_clip_dir& d = dirs[tn[t]];
    int32_t splits = 0, n_pad = 0, kp = 0;
    {
      int plan_splits = 0;
      int64_t need = 0;
      int rc = mmk_wgrad_plan(d.c, round_up(d.r, 128), k_pad, &plan_splits, &need);   // before anything is launched into tn_ws
      if (rc) return rc;
      MMK_REQUIRE(need <= d.tn_ws_floats, "tn_ws too small (mmk_wgrad_plan(c, round_up(r, 128), k_pad))");
    }
    int rc = mmk_wgrad_partial(d.g, d.y, d.tn_ws, d.c, round_up(d.r, 128), k_pad, d.ldg, k_pad, &splits, &n_pad, &kp, st);
    if (rc) return rc;
    fb.p[tn[t]] = FinProb{d.tn_ws, (long)n_pad * kp, kp, d.r, d.kappa, d.dx, d.dx_rows, d.dx_accumulate, d.src, d.normalize, splits};
    ld_max = std::max(ld_max, (int)kp);
  }
  {
    int rc = launch_grad_finalize(fb, n_dirs, max_r, ld_max, scale, upstream, db, dscale_out, dirs[0].dx_dtype, st);
    if (rc) return rc;
  }
  return 0;
}

int launch_grad_finalize(const FinBatch& fb, int n_dirs, int max_r, int ld_max, const float* scale, const float* upstream, const DsBatch& db,
                         float* dscale_out, int dx_dtype, hipStream_t st) {
  // dispatch-stamped events (hipExtLaunchKernelGGL), like the MFMA kernels of the path: a record-before / record-after pair reads
  // the queue wait of a short kernel behind a long one as its duration (168 us for a 5 us launch in the round-3 driver line)
  ProfEvents pe(MMK_K_GRAD_FINALIZE);
  int rc = MMK_DISPATCH_DTYPE(dx_dtype, U, [&]() -> int {
    hipExtLaunchKernelGGL((grad_finalize_kernel<U>), dim3(cdiv(max_r, 4), n_dirs + (dscale_out ? 1 : 0)), dim3(256), 4 * ld_max * sizeof(float), st,
                          pe.start, pe.stop, 0, fb, scale, upstream, db, dscale_out, n_dirs);
    return 0;
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

}  // namespace mmk

using namespace mmk;

extern "C" {

int mmk_clip_mirror_tiles(int r) { return 2 * cdiv(r, 64); }

int mmk_clip_plan(int r, int c, int k_pad, int compute, int32_t* n_col_tiles, int32_t* n_grad_blocks, int32_t* n_split) {
  MMK_REQUIRE(r > 0 && c > 0 && k_pad > 0, "empty problem");
  // sized for the worst case over tile choices / batch sizes so that callers need not know them
  if (n_col_tiles) *n_col_tiles = cdiv(c, 64);
  if (n_grad_blocks) *n_grad_blocks = (round_up(c, 128) / 64) * (round_up(r, 128) / 64);
  int split = 1;
  for (int nd = 1; nd <= MAX_PROBS; ++nd) split = std::max(split, make_plan(r, c, k_pad, nd, compute).n_split);
  if (n_split) *n_split = split;
  return 0;
}

int mmk_pack_rows(const void* src, int src_dtype, int n_src, int d, const int32_t* idx, int r, int normalize, void* dst,
                  void* dstT, int r_pad, int k_pad, int ldt, int compute, float* norm_out, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int bk = compute == MMK_COMPUTE_BF16 ? 64 : 32;
  MMK_REQUIRE(src && dst, "null pointer");
  MMK_REQUIRE(d > 0 && r >= 0 && n_src >= 0, "bad shape");
  MMK_REQUIRE(r_pad % 128 == 0 && r_pad >= r && r_pad > 0, "r_pad must be a positive multiple of 128 and >= r");
  MMK_REQUIRE(k_pad % bk == 0 && k_pad >= d && k_pad % 64 == 0, "k_pad must be a multiple of 64 and >= d");
  MMK_REQUIRE(dstT == nullptr || ldt == r_pad, "ldt must equal r_pad");
  {
    ProfScope ps(MMK_K_PACK, st);
    int rc = MMK_DISPATCH_DTYPE(src_dtype, S, [&]() -> int {
      if (compute == MMK_COMPUTE_BF16)
        hipLaunchKernelGGL((pack_rows_kernel<S, bf16_t>), dim3(r_pad / 4), dim3(256), 0, st, static_cast<const S*>(src), d,
                           idx, r, normalize, static_cast<bf16_t*>(dst), r_pad, k_pad, norm_out);
      else
        hipLaunchKernelGGL((pack_rows_kernel<S, float>), dim3(r_pad / 4), dim3(256), 0, st, static_cast<const S*>(src), d,
                           idx, r, normalize, static_cast<float*>(dst), r_pad, k_pad, norm_out);
      return 0;
    });
    if (rc) return rc;
    MMK_LAUNCH_CHECK();
  }
  if (dstT) {
    ProfScope ps(MMK_K_TRANSPOSE, st);
    dim3 grid(k_pad / 64, r_pad / 64);
    if (compute == MMK_COMPUTE_BF16)
      hipLaunchKernelGGL((transpose_kernel<bf16_t>), grid, dim3(256), 0, st, static_cast<const bf16_t*>(dst),
                         static_cast<bf16_t*>(dstT), r_pad, k_pad, ldt);
    else
      hipLaunchKernelGGL((transpose_kernel<float>), grid, dim3(256), 0, st, static_cast<const float*>(dst),
                         static_cast<float*>(dstT), r_pad, k_pad, ldt);
    MMK_LAUNCH_CHECK();
  }
  return 0;
}

int mmk_pack_rows_many(const mmk_pack_req* reqs, int n, int src_dtype, int d, int k_pad, int compute, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int bk = compute == MMK_COMPUTE_BF16 ? 64 : 32;
  MMK_REQUIRE(reqs && n > 0 && n <= MAX_PROBS, "1..8 operands per call");
  MMK_REQUIRE(d > 0 && k_pad % bk == 0 && k_pad >= d && k_pad % 64 == 0, "k_pad must be a multiple of 64 and >= d");
  PackBatch b;
  int r_pad_max = 0;
  for (int k = 0; k < n; ++k) {
    const mmk_pack_req& q = reqs[k];
    MMK_REQUIRE(q.src && q.dst && q.r >= 0, "null pointer / bad shape");
    MMK_REQUIRE(q.r_pad % 128 == 0 && q.r_pad >= q.r && q.r_pad > 0, "r_pad must be a positive multiple of 128 and >= r");
    MMK_REQUIRE(q.dstT == nullptr || q.ldt == q.r_pad, "ldt must equal r_pad");
    b.e[k] = PackEntry{q.src, q.idx, q.dst, q.dstT, q.r, q.r_pad, q.normalize, q.ldt, q.norm};
    r_pad_max = std::max(r_pad_max, q.r_pad);
  }
  ProfEvents pe(MMK_K_PACK);   // dispatch-stamped (see launch_grad_finalize)
  int rc = MMK_DISPATCH_DTYPE(src_dtype, S, [&]() -> int {
    if (compute == MMK_COMPUTE_BF16)
      hipExtLaunchKernelGGL((pack_tr_kernel<S, bf16_t>), dim3(r_pad_max / 16, n), dim3(256), 0, st, pe.start, pe.stop, 0, b, d, k_pad);
    else
      hipExtLaunchKernelGGL((pack_tr_kernel<S, float>), dim3(r_pad_max / 16, n), dim3(256), 0, st, pe.start, pe.stop, 0, b, d, k_pad);
    return 0;
  });
  if (rc) return rc;
  MMK_LAUNCH_CHECK();
  return 0;
}

static int check_dirs(const mmk_clip_dir* dirs, int n_dirs, int k_pad, int compute) {
  MMK_REQUIRE(dirs && n_dirs > 0 && n_dirs <= MAX_PROBS, "1..8 directions per call");
  const int bk = compute == MMK_COMPUTE_BF16 ? 64 : 32;
  MMK_REQUIRE(k_pad > 0 && k_pad % 64 == 0 && k_pad % bk == 0, "k_pad must be a positive multiple of 64");
  for (int k = 0; k < n_dirs; ++k) {
    const mmk_clip_dir& d = dirs[k];
    MMK_REQUIRE(d.x && d.y, "null operand");
    MMK_REQUIRE(d.r > 0 && d.c > 0, "empty direction");
    MMK_REQUIRE(d.label_off >= 0 && d.label_off + d.r <= d.c, "labels / row offset out of range");
    MMK_REQUIRE(d.mode == dirs[0].mode && (d.mode == 0 || d.mode == 1), "all directions of one call must share mode 0 or 1");
    MMK_REQUIRE(d.mode == 0 || d.hmax != nullptr, "alignment mode needs hmax");
  }
  return 0;
}

// 4-byte slots of the workspace mmk_clip_forward_loss needs: [0] the ticket counter (zero on entry, zero on exit), then one
// float per merge workgroup (their contents need no initialisation)
int mmk_clip_tickets(const mmk_clip_dir* dirs, int n_dirs) {
  int rows = 1, n = 0;
  for (int k = 0; k < n_dirs; ++k) {
    rows = std::max(rows, std::max(dirs[k].r, dirs[k].mirror_part != nullptr ? dirs[k].c : 0));
    n += dirs[k].mirror_part != nullptr ? 2 : 1;
  }
  return 1 + n * cdiv(rows, 64);
}

int mmk_clip_forward_loss(const mmk_clip_dir* dirs, int n_dirs, int k_pad, int d, int compute, const float* scale, const float* loss_w,
                          float* loss_out, int32_t* tickets, int n_tickets, void* stream) {
  (void)d;
  int rc = check_dirs(dirs, n_dirs, k_pad, compute);
  if (rc) return rc;
  MMK_REQUIRE(scale, "null scale");
  MMK_REQUIRE((loss_out == nullptr) == (loss_w == nullptr), "loss_w and loss_out come together");
  MMK_REQUIRE(loss_out == nullptr || dirs[0].mode == 0, "the in-launch loss value is for cross-entropy directions");
  for (int k = 0; k < n_dirs; ++k) {
    MMK_REQUIRE(dirs[k].part && dirs[k].loss_part && (dirs[k].mode == 1 || (dirs[k].diag && dirs[k].lse)), "null forward buffer");
    if (dirs[k].mirror_part != nullptr)
      MMK_REQUIRE(dirs[k].mode == 0 && dirs[k].r == dirs[k].c && dirs[k].label_off == 0 && dirs[k].mirror_lse && dirs[k].mirror_loss_part,
                  "a mirrored direction needs mode 0, r == c, label_off == 0 and its output buffers");
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (compute == MMK_COMPUTE_BF16) return clip_forward_impl<bf16_t>(dirs, n_dirs, k_pad, scale, loss_w, loss_out, tickets, n_tickets, st);
  return clip_forward_impl<float>(dirs, n_dirs, k_pad, scale, loss_w, loss_out, tickets, n_tickets, st);
}

int mmk_clip_forward(const mmk_clip_dir* dirs, int n_dirs, int k_pad, int d, int compute, const float* scale, int32_t* tickets,
                     int n_tickets, void* stream) {
  return mmk_clip_forward_loss(dirs, n_dirs, k_pad, d, compute, scale, nullptr, nullptr, tickets, n_tickets, stream);
}

int mmk_reduce_sums(const float* const* ptrs, const int32_t* counts, const float* weights, int n, int separate, float* out,
                    void* stream) {
  MMK_REQUIRE(ptrs && counts && weights && out && n > 0 && n <= 2 * MAX_PROBS, "bad arguments");
  CombineArgs a;
  a.n = n;
  a.separate = separate;
  for (int k = 0; k < n; ++k) {
    a.ptr[k] = ptrs[k];
    a.cnt[k] = counts[k];
    a.w[k] = weights[k];
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_LOSS_COMBINE, st);
  hipLaunchKernelGGL(reduce_sums_kernel, dim3(1), dim3(64), 0, st, a, out);
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_clip_backward(const mmk_clip_dir* dirs, int n_dirs, int k_pad, int d, int compute, const float* scale,
                      const float* upstream, float* dscale_out, void* stream) {
  int rc = check_dirs(dirs, n_dirs, k_pad, compute);
  if (rc) return rc;
  MMK_REQUIRE(scale && upstream, "null scale/upstream");
  MMK_REQUIRE(d > 0 && d <= k_pad, "bad d");
  const int dt = dirs[0].dx_dtype;
  const BwdFusedPlan fp = bwd_fused_plan(dirs, n_dirs, k_pad, compute);
  for (int k = 0; k < n_dirs; ++k) {
    const mmk_clip_dir& q = dirs[k];
    if (fp.fused[k])   // one kernel: neither the transposed operand nor G is touched
      MMK_REQUIRE(q.slab && q.lse && q.ds_part && q.dx, "null backward buffer");
    else
      MMK_REQUIRE((q.g_transposed || (q.yT && q.slab)) && (q.lse || q.mode == 1) && q.g && q.ds_part && q.dx,
                  "null backward buffer (yT and g may only be omitted for the directions mmk_clip_backward_plan marks fused)");
    MMK_REQUIRE(q.dx_dtype == dt, "all directions of one call must share dx_dtype");
    MMK_REQUIRE(!q.dx_accumulate || q.dx_dtype == MMK_F32, "accumulating scatter needs an f32 gradient buffer");
    MMK_REQUIRE(!q.normalize || (q.src && q.src_dtype == q.dx_dtype), "normalize backward needs src of dx dtype");
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  return MMK_DISPATCH_DTYPE(dt, U, [&]() -> int {
    if (compute == MMK_COMPUTE_BF16)
      return clip_backward_impl<bf16_t, U>(dirs, n_dirs, k_pad, d, scale, upstream, dscale_out, st);
    return clip_backward_impl<float, U>(dirs, n_dirs, k_pad, d, scale, upstream, dscale_out, st);
  });
}

int mmk_clip_backward_plan(const mmk_clip_dir* dirs, int n_dirs, int k_pad, int compute, int32_t* fused) {
  MMK_REQUIRE(dirs && fused && n_dirs >= 1 && n_dirs <= MAX_PROBS, "bad direction list");
  MMK_REQUIRE(compute == MMK_COMPUTE_BF16 || compute == MMK_COMPUTE_F32, "bad compute");
  const BwdFusedPlan fp = bwd_fused_plan(dirs, n_dirs, k_pad, compute);
  for (int k = 0; k < n_dirs; ++k) fused[k] = fp.fused[k] ? 1 : 0;
  return 0;
}

}  // extern "C"
