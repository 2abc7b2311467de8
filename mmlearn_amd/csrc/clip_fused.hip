// One-launch CLIP / InfoNCE loss WITH its gradients for small batches (every pair <= 1024 matched rows, one rank).
//
// Replaces, per LossPairSpec, the whole ATen sequence of mmlearn/modules/losses/contrastive.py:279-291 (emb[indices]),
// :327-340 (_safe_matmul, logit_scale *), :134-144 (F.cross_entropy x2, /2 * weight) and its autograd.  At N = 1024 the tiled
// multi-launch path (clip.hip) is launch-bound: seven launches of 4.5-11 us for 3.2 GFLOP of work.  Here ONE resident grid
// does all of it; workgroups hand data to each other through L2 with the write-through (sc1) store / agent-scope counter /
// sc1 load protocol of the CDNA4 guide (Guideline 16, table row "one lane of each storing workgroup adds to a counter"):
//
//   phase 1  workgroup (ti, tj): the 64 x 64 tile  S = B[tj] A[ti]^T  (bf16 MFMA 32x32x16, f32 accumulate) straight from the
//            user's rows (f32 or bf16, gathered through the match index, converted while staging -- no pack launch), per-row
//            and per-column (max, sum 2^(u - max)) of the tile -> L2; counters c1_row[ti], c1_col[tj].
//   phase 2  waits for its row strip's and column strip's 2 * nt partials, merges them to the row / column log-sum-exps, forms
//            G = P_row + P_col - 2 delta from the accumulators it still holds (no recompute), stores its tile of G (bf16,
//            2 MB at N = 1024: it never leaves the caches), the tile's share of d loss / d scale; the diagonal workgroups add
//            the loss terms; counters c2_row[ti], c2_col[tj].
//   phase 3  workgroup w takes output tiles of  dA = G B  and  dB = G^T A  (64 rows x 64 of the D columns, contraction over
//            all N).  dB: out[n][k] = sum_m G[m][n] A[m][k], both operands with the contraction along their rows -> LDS images
//            read with ds_read_b64_tr_b16.  dA: the same G is the MFMA A operand as it lies in memory (its rows are the output
//            rows), only B is read transposed -- G^T is never stored.  Raw f32 sums go to the workspace.
//   tail     the workgroup that draws the last ticket adds the loss / d-scale partials in a fixed order and re-arms every
//            counter (all zero on entry, all zero on exit).
// The backward call is one launch of the existing finalize kernel (x kappa * scale * upstream, cast, scatter).
//
// Why G goes through L2 instead of f32 atomics into the gradients: every (ti, tj) tile contributes a 64 x D partial to dA[ti]
// and to dB[tj], 16 adders per element at N = 1024 -- 64 MB of atomic adds against the part's ~1.3 TB/s atomic rate is ~50 us;
// G itself is 2 MB.
//
// Residency: a workgroup spins on counters that other workgroups of the same launch advance, so the whole grid must be
// co-resident: mmk_clip_fused_plan reports the capacity (occupancy query x CUs, at most 2 workgroups per CU) and the host
// only takes this path when the grid fits.  Every spin is bounded (s_memrealtime); a timeout poisons the loss with NaN.
#include <mutex>
#include <hip/hip_ext.h>

#include <algorithm>

#include "clip_internal.h"
#include "common.h"

namespace mmk {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));

constexpr int FT = 64;                 // tile edge
constexpr int F_NT_MAX = 16;           // <= 1024 matched rows per pair
constexpr int F_THREADS = 256;
constexpr int F_MAX_PAIRS = 4;       // two gradient directions per pair, MAX_PROBS directions per finalize launch
constexpr int F_KS = 128;              // phase 1: K elements per LDS stage (two 64-element sub-images)
constexpr int F_STAGE = 32 * 1024;     // bytes of one LDS stage (both phases)
constexpr int F_SCRATCH = 6 * 1024;
constexpr int F_LDS = 2 * F_STAGE + F_SCRATCH;
constexpr int F_CNT_STRIDE = 16;       // counters on lines of their own (64 B)
constexpr unsigned long long F_SPIN_LIMIT = 400000000ull;   // 4 s of the 100 MHz realtime clock

struct FusedPair {
  const char* a;          // raw rows of modality a [*, d] (src dtype)
  const char* b;
  const int32_t* idx_a;   // matched-row index lists (null: identity)
  const int32_t* idx_b;
  int n, nt, n_pad;
  int tile0, job0;        // first tile / first phase-3 job of this pair
  float w;                // weight / (2 n): factor of the pair's loss sums and of d/dscale
  float2* part_row;       // [nt][n_pad]  (tile column tj, row i)
  float2* part_col;       // [nt][n_pad]  (tile row ti, column j)
  bf16_t* G;              // [n_pad][n_pad]  G[i][j]
  unsigned* cnt;          // [4][nt] counters, stride F_CNT_STRIDE: c1_row, c1_col, c2_row, c2_col
  float* loss_part;       // [2 nt]
  float* ds_part;         // [nt * nt]
  float* dA;              // [n_pad][k_pad] raw gradient sums
  float* dB;
};
struct FusedArgs {
  FusedPair p[MAX_PROBS];
  int n_pairs, n_tiles, n_jobs;
  int d, k_pad, nkc;
  int want_grad;
  const float* scale;
  unsigned* done;         // [0] ticket, [F_CNT_STRIDE] error flag
  float* loss_out;
  float* ds_out;
  int dbg;                      // -DMMK_DEBUG_SWITCHES builds (MMK_FUSED_DBG; WRONG results, timing only): 1 no U loads, 2 no V loads, 4 no
                                // transposed reads / MFMAs, 8 no LDS stores in the gradient loop, 16 no phase-1 loads, 64 no gradient stores
  unsigned long long* stamps;   // -DMMK_DEBUG_SWITCHES builds: [grid][16] realtime-clock stamps of every workgroup's phases, or null
};
// phase stamps (100 MHz clock): compiled out of the product build
#define F_STAMP(k)                                                                                                          \
  do {                                                                                                                      \
    if (kDebugSwitches && a.stamps != nullptr && threadIdx.x == 0 && (int)blockIdx.x < a.n_tiles)                                       \
      a.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime();                                                        \
  } while (0)

__device__ __forceinline__ unsigned ld_cnt(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void add_cnt(unsigned* p) { __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// one lane: wait until both counters have reached `target`; false on timeout (error word set)
__device__ __forceinline__ bool spin_until(const unsigned* c0, const unsigned* c1, unsigned target, unsigned* err) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    const unsigned v0 = ld_cnt(c0), v1 = ld_cnt(c1);
    if (v0 >= target && v1 >= target) return true;
    __builtin_amdgcn_s_sleep(4);
    if (__builtin_amdgcn_s_memrealtime() - t0 > F_SPIN_LIMIT) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
}

template <typename S>
struct SrcT;
template <>
struct SrcT<float> {
  static constexpr int EPP = 4;   // elements per 16-byte piece
  // 4 floats -> 4 bf16 in 8 bytes
  static __device__ __forceinline__ uint2 to_bf16(const uint4& v) {
    typedef bf16_t bf2 __attribute__((ext_vector_type(2)));
    bf2 lo, hi;
    lo[0] = (bf16_t)__uint_as_float(v.x); lo[1] = (bf16_t)__uint_as_float(v.y);
    hi[0] = (bf16_t)__uint_as_float(v.z); hi[1] = (bf16_t)__uint_as_float(v.w);
    return make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
  }
};
template <>
struct SrcT<bf16_t> {
  static constexpr int EPP = 8;
};

__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }   // row of accumulator register e

template <typename S>
__global__ __launch_bounds__(F_THREADS, 2) void clip_fused_kernel(const FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int EPP = SrcT<S>::EPP;
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;

  // ---- scratch carve (behind the two stages)
  char* scr = smem + 2 * F_STAGE;
  float2* red_row = reinterpret_cast<float2*>(scr);             // [2][64]
  float2* red_col = reinterpret_cast<float2*>(scr + 1024);      // [2][64]
  float* cmax_w = reinterpret_cast<float*>(scr + 2048) + wave * 32;   // [4][2][16]
  float* lr2_s = reinterpret_cast<float*>(scr + 2560);          // [64]
  float* lc2_s = reinterpret_cast<float*>(scr + 2816);          // [64]
  float* diag_s = reinterpret_cast<float*>(scr + 3072);         // [64]
  float* misc = reinterpret_cast<float*>(scr + 3328);           // [16]
  int* flag_s = reinterpret_cast<int*>(scr + 3392);             // [4]

  unsigned* err = a.done + F_CNT_STRIDE;

  const int dbg = kDebugSwitches ? a.dbg : 0;
  F_STAMP(0);
  // Workgroups beyond the tile count are HELPERS: the host launches max(tiles, gradient jobs) workgroups (within the co-resident
  // capacity), because several small pairs have many more gradient jobs than tiles -- three pairs of 256 rows: 48 tiles, 192
  // jobs -- and a grid of 48 ran four jobs back to back per workgroup on a fifth of the chip.  A helper skips phases 1 and 2.
  const bool helper = (int)blockIdx.x >= a.n_tiles;
  if (!helper) {
  // ---- which pair / tile
  // XCD-aware: block b runs on XCD b % 8 and every XCD has its own L2, so the blocks of one XCD take CONSECUTIVE positions of
  // the tile list, and the list walks each pair in super-tiles of 4 x 8 tiles: at nt = 16 an XCD's 32 workgroups need 4 row
  // strips of A and 8 of B (768 rows through that L2) instead of all 2048 (r03_pmc_traffic.json: FETCH 9.2 -> see DESIGN 5)
  const int bp = ((int)blockIdx.x & 7) * (a.n_tiles >> 3) + min((int)blockIdx.x & 7, a.n_tiles & 7) + ((int)blockIdx.x >> 3);
  int pi = 0;
#pragma unroll 1
  while (pi + 1 < a.n_pairs && bp >= a.p[pi + 1].tile0) ++pi;
  const FusedPair& p = a.p[pi];
  const int nt = p.nt, n = p.n, n_pad = p.n_pad;
  int ti, tj;
  {
    const int t = bp - p.tile0;
    const int sr = t / (4 * nt), t2 = t - sr * 4 * nt, rh = min(4, nt - 4 * sr);
    const int sc = t2 / (rh * 8), t3 = t2 - sc * rh * 8, cw = min(8, nt - 8 * sc);
    ti = 4 * sr + t3 / cw;
    tj = 8 * sc + t3 % cw;
  }
  const int tile = ti * nt + tj;
  const float s = *a.scale;
  const float s2 = s * LOG2E;
  unsigned* c1_row = p.cnt + (0 * nt + ti) * F_CNT_STRIDE;
  unsigned* c1_col = p.cnt + (1 * nt + tj) * F_CNT_STRIDE;
  unsigned* c2_row = p.cnt + (2 * nt + ti) * F_CNT_STRIDE;
  unsigned* c2_col = p.cnt + (3 * nt + tj) * F_CNT_STRIDE;

  // =============================================================== phase 1: S tile
  // stage image: two sub-images (k halves) of [128 rows][128 B]; rows 0..63 = B[tj] (-> accumulator registers: columns j of S),
  // rows 64..127 = A[ti] (-> lanes: rows i of S); 16-byte chunk ^= (row >> 1) & 7 (conflict-free ds_read_b128 fragments)
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  {
    constexpr int PPR = F_KS / EPP;                 // 16-byte pieces per row and stage
    constexpr int NP = 128 * PPR / F_THREADS;       // pieces per thread and stage (16 / 8)
    constexpr int RPS = F_THREADS / PPR;            // rows covered by one piece slot (8 / 16)
    static_assert(64 % RPS == 0, "a piece slot never straddles the B / A halves of the stage image");
    const int c = tid % PPR;                        // piece within the row: K elements c*EPP ..
    const int trow = (tid & (F_THREADS - 1)) / PPR; // 0 .. RPS-1
    long srow[NP];                                  // byte offset of the source row, -1: row beyond n (zeros)
    int lds_off[NP];
    // Which operand a piece slot belongs to is a compile-time fact (u * RPS < 64): the pointers below stay in scalar registers.
    // (With a per-thread `row < 64` the compiler fetched p.a / p.b / p.idx_* from the kernarg segment with per-lane global loads
    // and waited for each -- four dependent round trips before the first operand load, 1.4 us of the launch.)
    const char* const pa = p.a;
    const char* const pb = p.b;
    const int32_t* const ia = p.idx_a;
    const int32_t* const ib = p.idx_b;
    int src_row[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) src_row[u] = (u * RPS < 64 ? tj : ti) * 64 + ((u * RPS + trow) & 63);
    // gathered pairings: all index loads of a half in one batch (clamped, unconditional), one wait
    if (ib != nullptr) {
#pragma unroll
      for (int u = 0; u < NP; ++u)
        if (u * RPS < 64) src_row[u] = src_row[u] < n ? ib[min(src_row[u], n - 1)] : -1;
    }
    if (ia != nullptr) {
#pragma unroll
      for (int u = 0; u < NP; ++u)
        if (u * RPS >= 64) src_row[u] = src_row[u] < n ? ia[min(src_row[u], n - 1)] : -1;
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int pp = (u * RPS < 64 ? tj : ti) * 64 + ((u * RPS + trow) & 63);
      if (pp >= n) src_row[u] = -1;
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int row = u * RPS + trow;
      srow[u] = src_row[u] >= 0 ? (long)src_row[u] * a.d * (long)sizeof(S) : -1;
      const int e0 = (c * EPP) & 63, kh = (c * EPP) >> 6;
      // the second sub-image keeps row r at row r ^ 1: the 16 lanes that store one row's two k halves then fill BOTH 128-byte halves
      // of a bank window (the sub-images are 16 KiB apart, i.e. on the same banks: every store was 2-way conflicted)
      lds_off[u] = kh * 16384 + (row ^ kh) * 128 + ((((e0 >> 3) ^ ((row >> 1) & 7))) << 4) + (e0 & 7) * 2;
    }
    const int ns = (a.k_pad + F_KS - 1) / F_KS;
    // The loop is latency-bound (128 KiB per workgroup, L2-resident): keep PD stages of loads in flight in registers
    // (bf16 rows: all of D = 512 at once), one barrier per stage.
    constexpr int PD = EPP == 8 ? 4 : 2;
    uint4 stg[PD][NP];
    auto load = [&](int st, uint4 (&v)[NP]) {
      const int col = st * F_KS + c * EPP;
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if (srow[u] >= 0 && col < a.d && !(dbg & 16))
          v[u] = *reinterpret_cast<const uint4*>((u * RPS < 64 ? pb : pa) + srow[u] + (long)col * (long)sizeof(S));
      }
    };
    auto store = [&](char* buf, const uint4 (&v)[NP]) {
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        if constexpr (EPP == 4) *reinterpret_cast<uint2*>(buf + lds_off[u]) = SrcT<float>::to_bf16(v[u]);
        else *reinterpret_cast<uint4*>(buf + lds_off[u]) = v[u];
      }
    };
    const int p_off = (wm * 32 + r) * 128, p_sw = ((wm * 32 + r) >> 1) & 7;
    const int q_off = (64 + wn * 32 + r) * 128, q_sw = ((64 + wn * 32 + r) >> 1) & 7;
    auto compute = [&](const char* buf, int st) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        if (st * F_KS + kh * 64 >= a.k_pad) break;
        const char* img = buf + kh * 16384;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int ch = 2 * kk + h;
          const bf16x8 fa = *reinterpret_cast<const bf16x8*>(img + (p_off ^ (kh << 7)) + ((ch ^ p_sw) << 4));
          const bf16x8 fb = *reinterpret_cast<const bf16x8*>(img + (q_off ^ (kh << 7)) + ((ch ^ q_sw) << 4));
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
      }
    };
    F_STAMP(8);
#pragma unroll
    for (int q = 0; q < PD; ++q)
      if (q < ns) load(q, stg[q]);
    F_STAMP(9);
#pragma unroll 1
    for (int st0 = 0; st0 < ns; st0 += PD) {
#pragma unroll
      for (int q = 0; q < PD; ++q) {
        const int st = st0 + q;
        if (st < ns) {
          // stage st -> buffer st & 1 (last read by compute(st - 2), which every thread finished before the barrier of stage st - 1)
          store(smem + (st & 1) * F_STAGE, stg[q]);
          if (st == 0) F_STAMP(10);
          if (st + PD < ns) load(st + PD, stg[q]);
          __syncthreads();
          compute(smem + (st & 1) * F_STAGE, st);
        }
      }
    }
    __syncthreads();   // the stage buffers are reused below
  }

  F_STAMP(1);
  // ---- tile statistics, both directions (log2 domain: u = s2 * t)
  const int i_loc = wn * 32 + r, i_glob = ti * 64 + i_loc;
  const bool iv = i_glob < n;
  const int j_base = tj * 64 + wm * 32;
  {
    float umax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float u = (j_base + acc_row(e, h) < n) ? acc[e] * s2 : -INFINITY;
      umax = fmaxf(umax, u);
    }
    umax = fmaxf(umax, __shfl_xor(umax, 32));
    float sum = 0.f;
    if (umax > -INFINITY) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sum += (j_base + acc_row(e, h) < n) ? fast_exp2(fmaf(acc[e], s2, -umax)) : 0.f;
    }
    sum += __shfl_xor(sum, 32);
    if (h == 0) red_row[wm * 64 + i_loc] = make_float2(umax, sum);
    // columns: per accumulator register over the 32 lanes (rows i) of the half-wave
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = iv ? acc[e] * s2 : -INFINITY;
    const float cm = half_wave_transpose_reduce<16, true>(v, lane);   // lane L: column L & 15 of half h
    cmax_w[h * 16 + (lane & 15)] = cm;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float mref = cmax_w[h * 16 + e];
      v[e] = (iv && mref > -INFINITY) ? fast_exp2(fmaf(acc[e], s2, -mref)) : 0.f;
    }
    const float cs = half_wave_transpose_reduce<16, false>(v, lane);
    const int cidx = lane & 15;
    if ((lane & 16) == 0) red_col[wn * 64 + wm * 32 + acc_row(cidx, h)] = make_float2(cm, cs);
    if (ti == tj && wm == wn && iv) {   // positive logits of the diagonal tile
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (acc_row(e, h) == r) diag_s[i_loc] = s * acc[e];
    }
  }
  __syncthreads();
  if (tid < 128) {
    const bool is_row = tid < 64;
    const int t = tid & 63;
    const float2* red = is_row ? red_row : red_col;
    const float2 x0 = red[t], x1 = red[64 + t];
    const float mx = fmaxf(x0.x, x1.x);
    float l = 0.f;
    if (x0.x > -INFINITY) l += x0.y * fast_exp2(x0.x - mx);
    if (x1.x > -INFINITY) l += x1.y * fast_exp2(x1.x - mx);
    const int g = (is_row ? ti : tj) * 64 + t;
    if (g < n) st_agent((is_row ? p.part_row + (size_t)tj * n_pad : p.part_col + (size_t)ti * n_pad) + g, make_float2(mx, l));
  }
  drain_vm();
  __syncthreads();
  if (tid == 0) {
    add_cnt(c1_row);
    add_cnt(c1_col);
    F_STAMP(2);
    flag_s[0] = spin_until(c1_row, c1_col, (unsigned)nt, err) ? 1 : 0;
  }
  __syncthreads();
  F_STAMP(3);
  const bool ok1 = flag_s[0] != 0;

  // =============================================================== phase 2: LSEs, loss terms, G
  if (ok1) {
    if (tid < 128) {
      const bool is_row = tid < 64;
      const int t = tid & 63;
      const int g = (is_row ? ti : tj) * 64 + t;
      float l2 = 0.f, term = 0.f;
      if (g < n) {
        const float2* part = (is_row ? p.part_row : p.part_col) + g;
        float mx = -INFINITY, l = 0.f;
        float2 pv[F_NT_MAX];
#pragma unroll
        for (int q = 0; q < F_NT_MAX; ++q) pv[q] = q < nt ? ld_agent(part + (size_t)q * n_pad) : make_float2(-INFINITY, 0.f);
#pragma unroll
        for (int q = 0; q < F_NT_MAX; ++q)
          if (pv[q].x > -INFINITY) {
            const float nm = fmaxf(mx, pv[q].x);
            l = l * fast_exp2(mx - nm) + pv[q].y * fast_exp2(pv[q].x - nm);
            mx = nm;
          }
        l2 = mx + log2f(l);
        if (ti == tj) term = l2 * LN2 - diag_s[t];
      }
      (is_row ? lr2_s : lc2_s)[t] = l2;
      F_STAMP(11);
      if (ti == tj) {   // waves 0 and 1: the row / column direction's sum over this strip (fixed order: deterministic)
        term = wave_sum(term);
        if (t == 0) st_agent(p.loss_part + (is_row ? ti : nt + tj), term);
      }
    }
    __syncthreads();
    if (a.want_grad) {
      const float lr2 = lr2_s[i_loc];
      float ds_acc = 0.f;
      char* gs = smem;                 // G tile  [64 i][64 j] bf16, row stride 144 B
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float g4[4];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const int e = 4 * q + e4;
          const int j_loc = wm * 32 + acc_row(e, h);
          const float t = acc[e];
          const float u = t * s2;
          float g = fast_exp2(u - lr2) + fast_exp2(u - lc2_s[j_loc]);
          if (ti == tj && j_loc == i_loc) g -= 2.f;
          if (!(iv && (tj * 64 + j_loc < n))) g = 0.f;
          ds_acc = fmaf(g, t, ds_acc);
          g4[e4] = g;
        }
        Vec4<bf16_t>::store(reinterpret_cast<bf16_t*>(gs + i_loc * 144) + wm * 32 + 8 * q + 4 * h, make_float4(g4[0], g4[1], g4[2], g4[3]));
      }
      F_STAMP(15);
      ds_acc = wave_sum(ds_acc);
      if (lane == 0) misc[wave] = ds_acc;
      __syncthreads();
      {
        const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.G), 0, n_pad * n_pad * 2, 0x00020000);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int pid = tid + F_THREADS * u, row = pid >> 3, cc = pid & 7;
          const u32x4 v0 = *reinterpret_cast<const u32x4*>(gs + row * 144 + cc * 16);
          __builtin_amdgcn_raw_buffer_store_b128(v0, rg, ((ti * 64 + row) * n_pad + tj * 64 + cc * 8) * 2, 0, 16);    // sc1: write-through
        }
      }
      if (tid == 0) st_agent(p.ds_part + tile, misc[0] + misc[1] + misc[2] + misc[3]);
      F_STAMP(12);
      drain_vm();
      __syncthreads();
      if (tid == 0) {
        add_cnt(c2_row);
        add_cnt(c2_col);
      }
    }
  }

  }   // !helper
  F_STAMP(4);
  // =============================================================== phase 3: dA = G B, dB = G^T A
  if (a.want_grad) {
    // Every wave takes a quarter of the contraction rows m for the whole 64 x 64 output tile and streams its own U / V rows
    // through a wave-private LDS image (no workgroup barrier inside the loop, D3 chunks of loads in flight in registers: the
    // loop is latency-bound); the four partial tiles are summed through LDS at the end.
    // chunk image: U [32 m][64 n] bf16 (4 KiB) then V [32 m][64 k] (4 KiB); 128-byte rows, 16-byte chunk ^= ((row >> 1) & 1) << 2
    // (the four rows of a transposed-read block on four different 64-byte groups of the 256-byte bank window)
    constexpr int CR = 32;
    constexpr int VPR = 64 / EPP;                       // V pieces per row (16 / 8)
    constexpr int NVL = CR * VPR / 64;                  // V loads per lane and chunk (8 / 4)
    constexpr int VRS = 64 / VPR;                       // rows per V load slot (4 / 8)
    // chunks in flight.  (r03: a one-workgroup-per-CU build with 512 registers and all 8 chunks of a 1024-row pair in flight ran the
    // loop in 3.0 us instead of 5.8 but waited 1.7 us longer for its first chunk -- the phase moves 256 KB per CU at ~40 GB/s either
    // way; not kept)
    constexpr int D3 = EPP == 8 ? 3 : 2;
    char* wbuf = smem + wave * 16384;                   // 2 x 8 KiB chunk images, later this wave's partial output tile
    const int li = lane & 15, q4 = li >> 2, p4 = li & 3, g1 = (lane >> 4) & 1;
    int tr_off[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) tr_off[t] = (8 * h + q4) * 128 + ((((t ^ (q4 >> 1)) * 4 + 2 * g1 + (p4 >> 1))) << 4) + 8 * (p4 & 1);
    auto tr8 = [&](const char* ptr) {
      const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ptr));
      const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(ptr + 512));
      s8v f;
      f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
      f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
      return __builtin_bit_cast(bf16x8, f);
    };
    // XCD-aware job order (speed only: any placement is correct).  Workgroups are dealt round-robin over the 8 XCDs, and every
    // XCD's L2 has to pull what its workgroups read through the fabric -- which is what bounds this phase (every XCD pulling
    // all of G and G^T: 32 MB at N = 1024).  The jobs are ordered (pair, direction, strip, k chunk); XCD x takes the contiguous
    // range [x, x + 1) * n_jobs / 8 = all k chunks of a few strips of one direction: their G strips reach that L2 once.
    const int ngrp = min(8, (int)gridDim.x);
    const int grp = (int)blockIdx.x % ngrp, slot = (int)blockIdx.x / ngrp;
    const int per_grp = (a.n_jobs + ngrp - 1) / ngrp;
    const int spg = ((int)gridDim.x - grp + ngrp - 1) / ngrp;      // workgroups of this group
    bool first_job = true;
#pragma unroll 1
    for (int jt = slot; jt < per_grp; jt += spg) {
      const int job = grp * per_grp + jt;
      if (job >= a.n_jobs) break;
      int pj = 0;
#pragma unroll 1
      while (pj + 1 < a.n_pairs && job >= a.p[pj + 1].job0) ++pj;
      const FusedPair& pp = a.p[pj];
      const int jl = job - pp.job0;
      const int per_dir = pp.nt * a.nkc;
      const int dir = jl / per_dir, strip = (jl % per_dir) / a.nkc, kc = (jl % per_dir) % a.nkc;
      // dB = G^T A: U[m = i][n = j] = G, both operands with the contraction along their rows (transposed reads).  dA = G B takes
      // the SAME matrix the other way round: its output rows are G's rows and the contraction runs along them, i.e. G is the MFMA
      // A operand as it lies in memory (row fragments, no transposed read) -- G^T is never stored (r03: it was, 2 MB of
      // write-through stores per launch that every tile waited for).
      const bf16_t* U = pp.G;
      const bool u_rows = dir == 0;   // workgroup-uniform
      const char* Vsrc = dir == 0 ? pp.b : pp.a;
      const int32_t* Vidx = dir == 0 ? pp.idx_b : pp.idx_a;
      float* out = dir == 0 ? pp.dA : pp.dB;
      const unsigned* cw = pp.cnt + ((2 + dir) * pp.nt + strip) * F_CNT_STRIDE;
      const int npad = pp.n_pad, nn = pp.n;
      const int mw = npad / 4;                            // contraction rows of this wave (a multiple of 16)
      const int m_lo = wave * mw, m_hi = m_lo + mw;
      const int nch = (mw + CR - 1) / CR;
      const auto ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(U), 0, npad * npad * 2, 0x00020000);

      u32x4 ust[D3][4];
      uint4 vst[D3][NVL];
      const int vc = lane % VPR;
      const int vcol = kc * 64 + vc * EPP;
      // Addresses are formed once per job: one wave per SIMD runs this loop alone, so every instruction in it is exposed.
      // A chunk whose 32 rows all exist (the usual case) loads without per-row checks; the row list of a gathered operand is
      // the only per-row lookup left.
      const int m_stop = min(m_hi, nn);                                   // V rows beyond are zero (padding of the pair)
      // dB: rows m of G, 8 rows x 128 B per load;  dA: rows i of the strip, columns m: 16 rows x 64 B per load
      const int u_voff = u_rows ? ((strip * 64 + (lane >> 2)) * npad + m_lo + (lane & 3) * 8) * 2
                                : ((m_lo + (lane >> 3)) * npad + strip * 64 + (lane & 7) * 8) * 2;
      const char* v_ptr = Vsrc + ((long)(m_lo + lane / VPR) * a.d + vcol) * (long)sizeof(S);
      const bool v_col_ok = vcol < a.d;
      auto load_v = [&](int ch, uint4 (&v)[NVL]) {
        const int m0 = m_lo + ch * CR;
        if (Vidx == nullptr && m0 + CR <= m_stop) {                       // wave-uniform
          const char* q = v_ptr + (long)ch * CR * a.d * (long)sizeof(S);
#pragma unroll
          for (int u = 0; u < NVL; ++u) {
            v[u] = make_uint4(0u, 0u, 0u, 0u);
            if (v_col_ok && !(dbg & 2)) v[u] = *reinterpret_cast<const uint4*>(q + (long)u * VRS * a.d * (long)sizeof(S));
          }
          return;
        }
#pragma unroll
        for (int u = 0; u < NVL; ++u) {
          const int m = m0 + u * VRS + lane / VPR;
          v[u] = make_uint4(0u, 0u, 0u, 0u);
          if (m < m_stop && v_col_ok && !(dbg & 2)) {
            const long srow = (long)(Vidx ? Vidx[m] : m) * a.d;
            v[u] = *reinterpret_cast<const uint4*>(Vsrc + (srow + vcol) * (long)sizeof(S));
          }
        }
      };
      auto load_u = [&](int ch, u32x4 (&w)[4]) {
        const int m0 = m_lo + ch * CR;
        if (u_rows) {   // [64 i][32 m]: load u = rows 16 u + (lane >> 2); a chunk's upper 16 columns may belong to the next wave
          const int off = u_voff + ch * CR * 2;
          const bool ok = m0 + (lane & 3) * 8 < m_hi;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            w[u] = (u32x4){0u, 0u, 0u, 0u};
            if (ok && !(dbg & 1)) w[u] = __builtin_amdgcn_raw_buffer_load_b128(ru, off + u * 16 * npad * 2, 0, 16);   // sc1
          }
          return;
        }
        const int off = u_voff + ch * CR * npad * 2;
        if (m0 + CR <= m_hi) {                                             // wave-uniform
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            w[u] = (u32x4){0u, 0u, 0u, 0u};
            if (!(dbg & 1)) w[u] = __builtin_amdgcn_raw_buffer_load_b128(ru, off + u * 8 * npad * 2, 0, 16);   // sc1
          }
          return;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          w[u] = (u32x4){0u, 0u, 0u, 0u};
          if (m0 + u * 8 + (lane >> 3) < m_hi && !(dbg & 1)) w[u] = __builtin_amdgcn_raw_buffer_load_b128(ru, off + u * 8 * npad * 2, 0, 16);
        }
      };
      auto store_uv = [&](char* buf, const u32x4 (&w)[4], const uint4 (&v)[NVL]) {
        if (u_rows) {   // image [64 i][64 B]: chunk ^= (row >> 2) & 3 -- the 16 rows of a ds_read_b128 group on 16 different slots
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int row = u * 16 + (lane >> 2), cc = lane & 3;
            *reinterpret_cast<u32x4*>(buf + row * 64 + ((cc ^ ((row >> 2) & 3)) << 4)) = w[u];
          }
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int row = u * 8 + (lane >> 3), cc = lane & 7;
            *reinterpret_cast<u32x4*>(buf + row * 128 + ((cc ^ (((row >> 1) & 1) << 2)) << 4)) = w[u];
          }
        }
#pragma unroll
        for (int u = 0; u < NVL; ++u) {
          const int row = u * VRS + lane / VPR;
          const int e0 = vc * EPP;
          char* dst = buf + 4096 + row * 128 + (((e0 >> 3) ^ (((row >> 1) & 1) << 2)) << 4) + (e0 & 7) * 2;
          if constexpr (EPP == 4) *reinterpret_cast<uint2*>(dst) = SrcT<float>::to_bf16(v[u]);
          else *reinterpret_cast<uint4*>(dst) = v[u];
        }
      };
      f32x16 o[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[i][j][e] = 0.f;
      // V does not depend on the hand-off: in flight while the strip's G tiles arrive
#pragma unroll
      for (int q = 0; q < D3; ++q)
        if (q < nch) load_v(q, vst[q]);
      __syncthreads();                 // flag_s / the LDS of the previous job (or of phase 2) are free
      if (tid == 0) {
        flag_s[1] = spin_until(cw, cw, (unsigned)pp.nt, err) ? 1 : 0;
      }
      __syncthreads();
      const bool ok = flag_s[1] != 0;   // workgroup-uniform
      if (first_job) F_STAMP(5);
      if (ok) {
#pragma unroll
        for (int q = 0; q < D3; ++q)
          if (q < nch) load_u(q, ust[q]);
#pragma unroll 1
        for (int c0 = 0; c0 < nch; c0 += D3) {
#pragma unroll
          for (int q = 0; q < D3; ++q) {
            const int ch = c0 + q;
            if (ch < nch) {
              // LDS operations of one wave execute in issue order: the image of chunk ch - 2 has been read when this lands
              char* buf = wbuf + (ch & 1) * 8192;
              if (!(dbg & 8)) store_uv(buf, ust[q], vst[q]);
              if (ch == 0 && first_job) F_STAMP(13);
              if (ch + D3 < nch) {
                load_u(ch + D3, ust[q]);
                load_v(ch + D3, vst[q]);
              }
              if (!(dbg & 4))
#pragma unroll
              for (int ks = 0; ks < CR / 16; ++ks) {
                bf16x8 fa0, fa1;
                if (u_rows) {   // A[row i][k = m]: lane (r, h) takes 8 consecutive m of row 32 t + r
                  fa0 = *reinterpret_cast<const bf16x8*>(buf + r * 64 + (((2 * ks + h) ^ ((r >> 2) & 3)) << 4));
                  fa1 = *reinterpret_cast<const bf16x8*>(buf + (32 + r) * 64 + (((2 * ks + h) ^ ((r >> 2) & 3)) << 4));
                } else {
                  fa0 = tr8(buf + tr_off[0] + ks * 2048);
                  fa1 = tr8(buf + tr_off[1] + ks * 2048);
                }
                const bf16x8 fb0 = tr8(buf + 4096 + tr_off[0] + ks * 2048), fb1 = tr8(buf + 4096 + tr_off[1] + ks * 2048);
                o[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, o[0][0], 0, 0, 0);
                o[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, o[0][1], 0, 0, 0);
                o[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, o[1][0], 0, 0, 0);
                o[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, o[1][1], 0, 0, 0);
              }
            }
          }
        }
        // partial tile of this wave -> its own 16 KiB as [64 n][64 k] f32; then every thread sums the four partials of 16
        // consecutive k of one row and stores them
        if (first_job) F_STAMP(14);
        float* part = reinterpret_cast<float*>(wbuf);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) part[(32 * i + acc_row(e, h)) * 64 + 32 * j + r] = o[i][j][e];
        __syncthreads();
        {
          // 16 lanes per 256-byte row of the partial tiles (one bank window: conflict-free ds_read_b128, whole-row global stores);
          // four passes of 16 rows.  (A thread per quarter row made every read 4-way conflicted: profiles/r03_loss_pmc_sq.json.)
          const int c = (tid & 15) * 4;
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 16 + (tid >> 4);
            float4 t = *reinterpret_cast<const float4*>(smem + (row * 64 + c) * 4);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
              const float4 x = *reinterpret_cast<const float4*>(smem + w * 16384 + (row * 64 + c) * 4);
              t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w;
            }
            if (!(dbg & 64)) *reinterpret_cast<float4*>(out + (size_t)(strip * 64 + row) * a.k_pad + kc * 64 + c) = t;
          }
        }
      }
      first_job = false;
    }
  }

  F_STAMP(6);
  // =============================================================== tail: last arriver
  drain_vm();
  __syncthreads();
  if (tid == 0) flag_s[2] = (int)__hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  F_STAMP(7);
  if ((unsigned)flag_s[2] != gridDim.x - 1) return;
  if (wave == 0) {
    float loss = 0.f, ds = 0.f;
#pragma unroll 1
    for (int k = 0; k < a.n_pairs; ++k) {
      const FusedPair& q = a.p[k];
      float lp = 0.f, dp = 0.f;
      for (int x = lane; x < 2 * q.nt; x += 64) lp += ld_agent(q.loss_part + x);
      if (a.want_grad)
        for (int x = lane; x < q.nt * q.nt; x += 64) dp += ld_agent(q.ds_part + x);
      loss += q.w * wave_sum(lp);
      ds += q.w * wave_sum(dp);
      for (int x = lane; x < 4 * q.nt; x += 64) __hip_atomic_store(q.cnt + x * F_CNT_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) {
      const bool bad = ld_cnt(err) != 0;
      *a.loss_out = bad ? __builtin_nanf("") : loss;
      if (a.ds_out) {
        a.ds_out[0] = bad ? __builtin_nanf("") : ds;
        a.ds_out[1] = 0.f;   // the accumulator mmk_clip_fused_backward adds upstream * ds to (saves the caller a fill launch)
      }
      __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(err, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------ host side
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct FusedLayout {
  size_t part_row, part_col, G, cnt, loss_part, ds_part, dA, dB, end;
};
// workspace: [0, 256): ticket + error words; then per pair the regions below
static FusedLayout pair_layout(size_t off, int n, int k_pad) {
  const int nt = cdiv(n, FT), n_pad = nt * FT;
  FusedLayout L;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off = align_up(off + bytes, 256);
    return at;
  };
  L.cnt = take((size_t)4 * nt * F_CNT_STRIDE * 4);
  L.loss_part = take((size_t)2 * nt * 4);
  L.ds_part = take((size_t)nt * nt * 4);
  L.part_row = take((size_t)nt * n_pad * 8);
  L.part_col = take((size_t)nt * n_pad * 8);
  L.G = take((size_t)n_pad * n_pad * 2);
  L.dA = take((size_t)n_pad * k_pad * 4);
  L.dB = take((size_t)n_pad * k_pad * 4);
  L.end = off;
  return L;
}

// Co-resident workgroups of clip_fused_kernel<S> on the CURRENT device.  Cached per device id (the > 64 KiB LDS opt-in is a per-device
// attribute of the function, and the CU count is the device's), under a mutex: the first calls can come from the main thread and
// an autograd worker at once.
template <typename S>
static int fused_capacity(int* out) {
  constexpr int MAX_DEV = 64;
  static int cached[MAX_DEV];
  static bool known[MAX_DEV] = {};
  static std::mutex mu;
  int dev = 0;
  MMK_HIP(hipGetDevice(&dev));
  MMK_REQUIRE(dev >= 0 && dev < MAX_DEV, "fused loss: device id out of range");
  std::lock_guard<std::mutex> lock(mu);
  if (!known[dev]) {
    auto kern = clip_fused_kernel<S>;
    MMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS));
    int per_cu = 0;
    MMK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, F_THREADS, F_LDS));
    hipDeviceProp_t prop;
    MMK_HIP(hipGetDeviceProperties(&prop, dev));
    cached[dev] = std::min(per_cu, 2) * prop.multiProcessorCount;   // LDS admits two workgroups per CU; never count on more
    known[dev] = true;
  }
  *out = cached[dev];
  return 0;
}

static unsigned long long* g_fused_stamps = nullptr;

}  // namespace mmk

using namespace mmk;

extern "C" {

// measurement hook (debug-switch builds only): device buffer of [grid][8] 64-bit stamps the next launches fill
int mmk_clip_fused_debug_stamps(unsigned long long* device_buf) {
  MMK_REQUIRE(kDebugSwitches || device_buf == nullptr, "phase stamps need a -DMMK_DEBUG_SWITCHES build of the library");
  g_fused_stamps = device_buf;
  return 0;
}

int mmk_clip_fused_plan(const int32_t* n, int n_pairs, int d, int src_dtype, int64_t* ws_bytes, int32_t* grid, int32_t* capacity) {
  MMK_REQUIRE(n && n_pairs > 0 && n_pairs <= F_MAX_PAIRS && d > 0, "fused loss: 1..4 pairs per call");
  MMK_REQUIRE(src_dtype == MMK_F32 || src_dtype == MMK_BF16, "fused loss: f32 or bf16 embeddings");
  MMK_REQUIRE(d % (src_dtype == MMK_F32 ? 4 : 8) == 0, "fused loss: rows must be whole 16-byte pieces");
  const int k_pad = round_up(d, 64);
  size_t off = 256;
  int tiles = 0;
  for (int k = 0; k < n_pairs; ++k) {
    MMK_REQUIRE(n[k] > 0 && n[k] <= FT * F_NT_MAX, "fused loss: 1..1024 matched rows per pair");
    off = pair_layout(off, n[k], k_pad).end;
    const int nt = cdiv(n[k], FT);
    tiles += nt * nt;
  }
  if (ws_bytes) *ws_bytes = (int64_t)off;
  if (grid) *grid = tiles;
  if (capacity) {
    int cap = 0;
    int rc = src_dtype == MMK_F32 ? fused_capacity<float>(&cap) : fused_capacity<bf16_t>(&cap);
    if (rc) return rc;
    *capacity = cap;
  }
  return 0;
}

static int fused_fill(const mmk_fused_pair* pairs, int n_pairs, int d, void* ws, int64_t ws_bytes, FusedArgs* a) {
  const int k_pad = round_up(d, 64);
  char* base = static_cast<char*>(ws);
  size_t off = 256;
  int tiles = 0, jobs = 0;
  a->n_pairs = n_pairs;
  a->d = d;
  a->k_pad = k_pad;
  a->nkc = k_pad / 64;
  for (int k = 0; k < n_pairs; ++k) {
    const mmk_fused_pair& q = pairs[k];
    MMK_REQUIRE(q.a && q.b && q.n > 0 && q.n <= FT * F_NT_MAX, "fused loss: bad pair");
    const FusedLayout L = pair_layout(off, q.n, k_pad);
    off = L.end;
    FusedPair& p = a->p[k];
    p.a = static_cast<const char*>(q.a);
    p.b = static_cast<const char*>(q.b);
    p.idx_a = q.idx_a;
    p.idx_b = q.idx_b;
    p.n = q.n;
    p.nt = cdiv(q.n, FT);
    p.n_pad = p.nt * FT;
    p.tile0 = tiles;
    p.job0 = jobs;
    p.w = q.weight / (2.f * (float)q.n);
    p.part_row = reinterpret_cast<float2*>(base + L.part_row);
    p.part_col = reinterpret_cast<float2*>(base + L.part_col);
    p.G = reinterpret_cast<bf16_t*>(base + L.G);
    p.cnt = reinterpret_cast<unsigned*>(base + L.cnt);
    p.loss_part = reinterpret_cast<float*>(base + L.loss_part);
    p.ds_part = reinterpret_cast<float*>(base + L.ds_part);
    p.dA = reinterpret_cast<float*>(base + L.dA);
    p.dB = reinterpret_cast<float*>(base + L.dB);
    tiles += p.nt * p.nt;
    jobs += 2 * p.nt * a->nkc;
  }
  MMK_REQUIRE((int64_t)off <= ws_bytes, "fused loss: workspace too small (mmk_clip_fused_plan)");
  a->n_tiles = tiles;
  a->n_jobs = jobs;
  a->done = reinterpret_cast<unsigned*>(base);
  return 0;
}

int mmk_clip_fused_forward(const mmk_fused_pair* pairs, int n_pairs, int d, int src_dtype, const float* scale, void* ws, int64_t ws_bytes,
                           int want_grad, float* loss_out, float* ds_out, void* stream) {
  MMK_REQUIRE(pairs && n_pairs > 0 && n_pairs <= F_MAX_PAIRS && scale && ws && loss_out, "fused loss: bad arguments (1..4 pairs per call)");
  MMK_REQUIRE(src_dtype == MMK_F32 || src_dtype == MMK_BF16, "fused loss: f32 or bf16 embeddings");
  MMK_REQUIRE(d > 0 && d % (src_dtype == MMK_F32 ? 4 : 8) == 0, "fused loss: rows must be whole 16-byte pieces");
  FusedArgs a;
  int rc = fused_fill(pairs, n_pairs, d, ws, ws_bytes, &a);
  if (rc) return rc;
  a.want_grad = want_grad;
  a.scale = scale;
  a.loss_out = loss_out;
  a.ds_out = ds_out;
  a.stamps = kDebugSwitches ? g_fused_stamps : nullptr;
  a.dbg = MMK_DBG_ENV("MMK_FUSED_DBG") ? atoi(MMK_DBG_ENV("MMK_FUSED_DBG")) : 0;
  int cap = 0;
  rc = src_dtype == MMK_F32 ? fused_capacity<float>(&cap) : fused_capacity<bf16_t>(&cap);
  if (rc) return rc;
  MMK_REQUIRE(a.n_tiles <= cap, "fused loss: the grid would not be co-resident (more tiles than mmk_clip_fused_plan's capacity)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  {
    ProfEvents pe(MMK_K_CLIP_FUSED);
    // helpers for the gradient phase (see the kernel): as many workgroups as there are jobs, within what is co-resident
    const int grid = a.want_grad ? std::min(cap, std::max(a.n_tiles, a.n_jobs)) : a.n_tiles;
    if (src_dtype == MMK_F32)
      hipExtLaunchKernelGGL(clip_fused_kernel<float>, dim3(grid), dim3(F_THREADS), F_LDS, st, pe.start, pe.stop, 0, a);
    else
      hipExtLaunchKernelGGL(clip_fused_kernel<bf16_t>, dim3(grid), dim3(F_THREADS), F_LDS, st, pe.start, pe.stop, 0, a);
  }
  MMK_LAUNCH_CHECK();
  return 0;
}

int mmk_clip_fused_backward(const mmk_fused_pair* pairs, int n_pairs, int d, int dx_dtype, const float* scale, const float* upstream,
                            void* ws, int64_t ws_bytes, const float* ds_raw, float* dscale_out, void* stream) {
  MMK_REQUIRE(pairs && n_pairs > 0 && n_pairs <= F_MAX_PAIRS && scale && upstream && ws, "fused loss: bad arguments (1..4 pairs per call)");
  FusedArgs a;
  int rc = fused_fill(pairs, n_pairs, d, ws, ws_bytes, &a);
  if (rc) return rc;
  FinBatch fb;
  DsBatch db;
  fb.d = d;
  int max_r = 0, n_dirs = 0;
  for (int k = 0; k < n_pairs; ++k) {
    const mmk_fused_pair& q = pairs[k];
    const float kappa = q.weight / (2.f * (float)q.n);
    MMK_REQUIRE(q.da && q.db, "fused loss: null gradient buffer");
    MMK_REQUIRE((!q.da_accumulate && !q.db_accumulate) || dx_dtype == MMK_F32, "accumulating scatter needs f32 gradient buffers");
    // Directions of this launch that add into the same identity-paired destination (two pairs sharing a modality) become ONE
    // problem with several sources: every element then has one writer here, so the accumulation into what earlier launches left
    // is a plain load + add + store instead of f32 atomics (3 pairs x 256 rows: 15.7 -> see DESIGN 3.1a).
    const float* raw[2] = {a.p[k].dA, a.p[k].dB};
    void* dst[2] = {q.da, q.db};
    const int32_t* rows[2] = {q.idx_a, q.idx_b};
    const int accf[2] = {q.da_accumulate, q.db_accumulate};
    for (int side = 0; side < 2; ++side) {
      int host = -1;
      if (accf[side] && rows[side] == nullptr)
        for (int j = 0; j < n_dirs; ++j)
          if (fb.p[j].dx == dst[side] && fb.p[j].dx_rows == nullptr && fb.p[j].accumulate && fb.p[j].r == q.n && fb.p[j].n_extra < 3) host = j;
      if (host >= 0) {
        FinProb& h = fb.p[host];
        h.extra[h.n_extra] = raw[side];
        h.kappa_extra[h.n_extra] = kappa;
        ++h.n_extra;
      } else {
        FinProb f{raw[side], 0, a.k_pad, q.n, kappa, dst[side], rows[side], accf[side], nullptr, 0, 1};
        f.exclusive = accf[side] && rows[side] == nullptr;   // identity pairing: row i is written by this problem's row i only ...
        fb.p[n_dirs++] = f;
      }
    }
    max_r = std::max(max_r, (int)q.n);
  }
  // ... unless another problem of this launch targets the same buffer with other rows (a gathered pairing): then both keep atomics
  for (int i = 0; i < n_dirs; ++i)
    for (int j = 0; j < n_dirs; ++j)
      if (i != j && fb.p[i].dx == fb.p[j].dx) fb.p[i].exclusive = 0;
  db.n_probs = 0;
  if (dscale_out) {
    MMK_REQUIRE(ds_raw, "fused loss: ds_raw required with dscale_out");
    db.part[0] = ds_raw;
    db.n[0] = 1;
    db.kappa[0] = 1.f;
    db.n_probs = 1;
  }
  return launch_grad_finalize(fb, n_dirs, max_r, a.k_pad, scale, upstream, db, dscale_out, dx_dtype, static_cast<hipStream_t>(stream));
}

}  // extern "C"
