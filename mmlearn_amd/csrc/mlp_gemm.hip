// the two GEMMs of the encoders' MLPs that sit next to an activation, each with the activation pass folded
// into its epilogue (mmlearn/modules/layers/mlp.py; HF CLIPMLP / BertIntermediate + BertOutput under
// mmlearn/modules/encoders/clip.py:29-470, text.py:20-178):
//
//   forward   H = act(X W1^T + b1), and the pre-activation X W1^T for the backward          (mode MG_FWD_ACT)
//   backward  dPre = (dY W2) * act'(Pre + b1),  db1 = column sums of dPre                    (mode MG_BWD_DACT)
//
// and the pair the product runs, in which the forward leaves the activation's DERIVATIVE behind instead of the pre-activation (the
// backward needs nothing else of it; same bytes) and the backward's epilogue is a multiply -- the exp / rcp chain of act' costs
// the forward three more instructions per element next to the ones act needs anyway, and cost the backward 160 us of a 1320 us
// launch at M = 201,728, exposed (an epilogue cannot hide behind MFMAs when one workgroup owns the CU):
//
//   forward   H = act(X W1^T + b1),  G = act'(X W1^T + b1)                                   (mode MG_FWD_ACT_G)
//   backward  dPre = (dY W2) * G,    db1 = column sums of dPre                               (mode MG_BWD_MUL)
//
//     C[M, N] = A[M, K] . B[N, K]^T       bf16 operands (rows K-contiguous), f32 accumulation, bf16 out
//
// The unfused step runs  library GEMM -> bias_act kernel : the backward pair writes dAct [M, 3072] (1.24 GB at M = 201,728),
// reads it back together with Pre and writes dPre -- 3.7 GB of HBM traffic next to a GEMM whose own operands are 0.3 GB.  Here
// the tile of Pre is read while the accumulators are still in registers and only dPre is written (2.8 GB in all).
//
// Main loop (round 2's second GEMM design, tools/probes/gemm4.hip, kept because it is the simplest loop that reaches 0.9 x the
// tuned library on these shapes): persistent workgroups walk 256 x 256 output tiles in an XCD-aware order; eight waves as
// 2 (m) x 4 (n), 128 x 64 per wave = 8 accumulator tiles of v_mfma_f32_32x32x16_bf16; operands arrive by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a ring of ten 16-KiB sub-slots (128 rows x 64 k), the XOR
// swizzle chunk ^= (row >> 1) & 7 applied on the per-lane SOURCE address so that ds_read_b128 fragment reads are conflict
// free; ONE barrier per K step with a counted s_waitcnt vmcnt (the next step's pieces stay in flight across it).
// The weight rows are the MFMA A operand and the activation rows the B operand, so an accumulator register group holds four
// CONSECUTIVE output columns of one row (8 bytes of bf16); the epilogue regroups them into whole 128-byte lines through a
// wave-private LDS image, so that every global access of the tile (C stores, Pre loads) is 8 rows x 128 B per wave-instruction.
#include <hip/hip_ext.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace mmk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int MG_TILE = 256;          // output tile edge
constexpr int MG_BK = 64;             // K per step
constexpr int MG_SUB = 128 * 128;     // bytes of one sub-slot: 128 rows x 64 bf16
constexpr int MG_RING = 10;           // sub-slots in the ring
constexpr int MG_LDS = MG_RING * MG_SUB;

enum { MG_PLAIN = 0, MG_FWD_ACT = 1, MG_BWD_DACT = 2, MG_FWD_ACT_G = 3, MG_BWD_MUL = 4 };
enum { MG_ACT_QUICK_GELU = 0, MG_ACT_GELU = 1 };   // numbering of mmk_bias_act_*

struct MlpGemmArgs {
  const bf16_t* A;    // [M, K] row stride lda
  const bf16_t* B;    // [N, K] row stride ldb
  bf16_t* C;          // [M, N] row stride ldc
  bf16_t* C2;         // MG_FWD_ACT: second output, the pre-activation A B^T without the bias; MG_FWD_ACT_G: act'(A B^T + bias) (stride ldc)
  const bf16_t* P;    // MG_BWD_DACT: pre-activation [M, N]; MG_BWD_MUL: the factor G [M, N]; row stride ldp
  const float* bias;  // [N] or null
  float* part;        // MG_BWD_DACT: column-sum partials f32[2 * tiles_m][N] (one row per 128 output rows), or null
  long lda, ldb, ldc, ldp;
  int M, N, K;
  int tiles_m, tiles_n;
  unsigned long long* stamps;   // debug-switch builds: shader-clock stamps of workgroup 0 (tools/mlp_gemm_stamps.py), else null
  int dbg;            // timing ablations, debug-switch builds only: 4 = no C stores, 8 = no epilogue arithmetic, 64 = hot DMA, 512 = no PIPE
};

__device__ __forceinline__ void mg_dma16(const void* sbase, uint32_t voff, uint32_t lds_addr) {
  const uint64_t pb = reinterpret_cast<uint64_t>(sbase);
  const uint64_t ps = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pb >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pb);
  lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  // no "memory" clobber: the DMA lands in ring slots nobody reads during this step (the barriers order it), and the compiler
  // must stay free to move this step's fragment reads across it
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(reinterpret_cast<const void*>(ps)), "s"(lds_addr));
}

__device__ __forceinline__ unsigned long long mg_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

template <int N>
__device__ __forceinline__ void mg_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// pack two f32 into one dword of two bf16 (RNE, NaN-preserving: v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t mg_pk(float lo, float hi) {
  typedef bf16_t bf2 __attribute__((ext_vector_type(2)));
  bf2 v;
  v[0] = (bf16_t)lo;
  v[1] = (bf16_t)hi;
  return __builtin_bit_cast(uint32_t, v);
}

// Phi(z) (standard normal CDF) and exp(-z^2 / 2) with ONE exponential: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7,
// i.e. below f32 rounding of the surrounding arithmetic), whose exp(-x^2) factor at x = z / sqrt(2) is the density's.
__device__ __forceinline__ float mg_phi(float z, float& e) {
  const float az = fabsf(z) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(az, 0.3275911f, 1.f));
  e = __builtin_amdgcn_exp2f(z * z * -0.72134752044448170f);   // exp(-z^2 / 2)
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float half_erf = fmaf(p * t, -0.5f * e, 0.5f);          // erf(|z| / sqrt 2) / 2
  return 0.5f + copysignf(half_erf, z);
}
template <int ACT>
__device__ __forceinline__ float mg_act(float z) {
  if (ACT == MG_ACT_QUICK_GELU) return z * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(z * -2.4554669595930156f));
  float e;
  return z * mg_phi(z, e);
}
template <int ACT>
__device__ __forceinline__ float mg_act_grad(float z) {
  if (ACT == MG_ACT_QUICK_GELU) {
    const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(z * -2.4554669595930156f));   // sigmoid(1.702 z)
    const float ts = 1.702f * z * sg;
    return fmaf(ts, 1.f - sg, sg);
  }
  float e;
  const float phi = mg_phi(z, e);
  return fmaf(z * 0.3989422804014327f, e, phi);
}

// act(z) and act'(z) from one exponential / reciprocal pair
template <int ACT>
__device__ __forceinline__ void mg_act_both(float z, float& y, float& g) {
  if (ACT == MG_ACT_QUICK_GELU) {
    const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(z * -2.4554669595930156f));   // sigmoid(1.702 z)
    y = z * sg;
    g = fmaf(1.702f * y, 1.f - sg, sg);
    return;
  }
  float e;
  const float phi = mg_phi(z, e);
  y = z * phi;
  g = fmaf(z * 0.3989422804014327f, e, phi);
}

// PIPE (MG_BWD_MUL, K >= 768): the tile's global I/O is software-pipelined against the K loops.  One CU moves ~10 bytes per cycle
// to or from HBM, so the 128 KiB of a tile's C stores and as much of G loads take ~5 us each, and vmcnt retires in order: a K
// step's closing s_waitcnt for its LDS-DMA also waits for every older store, and the loads of G stand in front of the epilogue.
// Here the epilogue only regroups the tile into 16 row-layout pieces per lane that stay in registers; the NEXT tile's first six K
// steps store them, two or three pieces per step (a step's closing wait then covers 24 KiB per CU, not 128), and the tile's last
// six K steps fetch its 16 pieces of G the same way.  Worth 60 us of 1190 at M = 201,728; nothing on the plain product.
template <int MODE, int ACT, bool PIPE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_gemm_kernel(const MlpGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // wave tile: output rows 128 wm .., output columns 64 wn ..
  const int r = lane & 31, h = lane >> 5;
  const uint32_t ring = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)smem);
  const int dbg = kDebugSwitches ? a.dbg : 0;

  // ---- this workgroup's tiles: XCD x = blockIdx % 8 takes tiles [(i * 8 + x) * per_xcd, + per_xcd), i = 0, 1, ...
  const int per_xcd = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int total = a.tiles_m * a.tiles_n;
  const int nk = a.K / MG_BK;
  int n_my = 0;
  for (int i = 0;; ++i) {
    if ((i * 8 + xcd) * per_xcd + slot >= total) break;
    ++n_my;
  }
  const int G = n_my * nk;
  if (G == 0) return;

  // ---- loader state.  Piece p = 2 * wave + u (u = 0, 1) of a sub-slot covers its rows 8p .. 8p + 7; lane L lands at LDS
  // row 8p + (L >> 3), chunk slot L & 7 and therefore fetches source chunk (L & 7) ^ swizzle(row).  M and N are multiples
  // of the tile (host check), so no row needs clamping and the per-lane byte offsets are loop constants; everything that
  // changes from step to step (tile, k position, half of the tile, ring slot) is scalar.
  uint32_t voffA[2], voffB[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int lrow = 8 * (2 * wave + u) + (lane >> 3);
    const uint32_t lchunk = (uint32_t)(((lane & 7) ^ ((lrow >> 1) & 7)) << 4);
    voffA[u] = (uint32_t)lrow * (uint32_t)(a.lda * 2) + lchunk;
    voffB[u] = (uint32_t)lrow * (uint32_t)(a.ldb * 2) + lchunk;
  }
  const uint32_t wave_lds = ring + (uint32_t)(2 * wave) * 1024u;
  // data cursors: the (tile, k position) whose A (B) sub-slots are issued next.  They stop at the last step: the ring schedule
  // below keeps issuing (the same data again, into slots nobody reads any more) so that the instruction counts the vmcnt
  // waits rely on never change and the K loop has no branch between its MFMAs.
  struct Cur {
    int step, kt, tm, tn;
  };
  // this workgroup's tiles are t0, t0 + dt, t0 + 2 dt, ...: (tm, tn) advance by (dm, dn) with a carry
  const int t0 = xcd * per_xcd + slot, dt = 8 * per_xcd;
  const int dm = dt / a.tiles_n, dn = dt - dm * a.tiles_n;
  auto next_tile = [&](int& tm, int& tn) {
    tm += dm;
    tn += dn;
    if (tn >= a.tiles_n) {
      tn -= a.tiles_n;
      ++tm;
    }
  };
  auto advance = [&](Cur& c) {
    if (c.step + 1 >= G) return;
    ++c.step;
    if (++c.kt == nk) {
      c.kt = 0;
      next_tile(c.tm, c.tn);
    }
  };
  Cur cA{0, 0, t0 / a.tiles_n, t0 % a.tiles_n};
  Cur cB = cA;
  int posA = 0, posB = 2;   // ring position of the next A / B pair of sub-slots: (4 s) mod 10, (4 s + 2) mod 10
  auto bump = [&](int& pos) {
    pos += 4;
    if (pos >= MG_RING) pos -= MG_RING;
  };
  auto wrap1 = [&](int pos) { return pos >= MG_RING ? pos - MG_RING : pos; };
  auto issue_sub = [&](const bf16_t* src, const uint32_t (&voff)[2], int pos) {   // src: first row of the 128-row half, k position applied
    const uint32_t dst = wave_lds + (uint32_t)pos * MG_SUB;
#pragma unroll
    for (int u = 0; u < 2; ++u) mg_dma16(src, voff[u], dst + (uint32_t)u * 1024u);
  };
  const bool hot = (dbg & 64) != 0;   // ablation: every DMA re-reads the same (cache-resident) 64 KiB
  auto srcA = [&](const Cur& c, int half) { return a.A + (hot ? 0 : ((long)c.tm * MG_TILE + 128 * half) * a.lda + (long)c.kt * MG_BK); };
  auto srcB = [&](const Cur& c, int half) { return a.B + (hot ? 0 : ((long)c.tn * MG_TILE + 128 * half) * a.ldb + (long)c.kt * MG_BK); };

  // ---- fragment read offsets inside a sub-slot: row (32 blk + r), chunk (2 kk + h) ^ ((r >> 1) & 7)
  uint32_t offk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) offk[kk] = (uint32_t)(r * 128 + (((2 * kk + h) ^ ((r >> 1) & 7)) << 4));

  // ---- prologue: A0 B0 A1.  Step g then issues the B half of step g + 1 and the A half of step g + 2 -- for g = 0 into the
  // four ring slots the prologue left empty, later into the slots step g - 1 released: the same eight instructions per wave
  // in every step.
  issue_sub(srcA(cA, 0), voffA, 0);
  issue_sub(srcA(cA, 1), voffA, 1);
  advance(cA);
  issue_sub(srcB(cB, 0), voffB, 2);
  issue_sub(srcB(cB, 1), voffB, 3);
  advance(cB);
  issue_sub(srcA(cA, 0), voffA, 4);
  issue_sub(srcA(cA, 1), voffA, 5);
  advance(cA);
  posA = 8;   // A of step 2
  posB = 6;   // B of step 1

  int ctm = t0 / a.tiles_n, ctn = t0 % a.tiles_n;   // compute cursor
  int p0 = 0;   // ring position of the step being computed: (4 g) mod 10
  // step 0 has landed when at most the A half of step 1 (4 instructions of this wave) is in flight
  mg_wait_vmcnt<4>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  u32x4 oq[16];               // PIPE: the previous tile's output in row layout (piece i: row 8 i + (lane >> 3), chunk lane & 7), not yet stored
  bf16_t* oq_ptr = nullptr;   // PIPE: this lane's address of piece 0 (piece i: + i * oq_step); null: nothing pending
  const long oq_step = 8 * a.ldc;
  for (int ti = 0; ti < n_my; ++ti) {
    f32x16 acc[2][4];   // [n block][m block]
    {
      f32x16 zero;
#pragma unroll
      for (int e = 0; e < 16; ++e) zero[e] = 0.f;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = zero;
    }
    int stamp_step = 0;
    constexpr bool HAS_P = MODE == MG_BWD_DACT || MODE == MG_BWD_MUL;
    u32x4 pr[PIPE ? 16 : 8];   // HAS_P: blocks of P in row layout (piece i: row 8 i + (lane >> 3), 16-byte chunk lane & 7)
    const bf16_t* pp0 = HAS_P ? a.P + (size_t)(ctm * MG_TILE + 128 * wm + (lane >> 3)) * a.ldp + ctn * MG_TILE + 64 * wn + 8 * (lane & 7) : nullptr;
    const long pstep = 8 * a.ldp;
    // One K step.  ST (PIPE): 0..5 = store slice ST of the previous tile's pending pieces first; LD (PIPE, HAS_P): 0..5 = request
    // slice LD of this tile's pieces of P first; LAST: the tile's last step -- its eight DMA instructions go out in the first two k
    // blocks and it closes on vmcnt(0) through the BUILTIN, which settles the compiler's own count of the loads of P (it cannot see
    // the LDS-DMA and would otherwise place vmcnt(15 .. 0) in front of the uses of P, draining that queue each time).
    auto k_step = [&](auto st_c, auto ld_c, auto last_c) {
      constexpr int ST = decltype(st_c)::value, LD = decltype(ld_c)::value;
      constexpr bool LAST = decltype(last_c)::value;
      constexpr int LO[7] = {0, 3, 6, 9, 12, 14, 16};
      if (PIPE && ST >= 0) {
        if (oq_ptr != nullptr) {
#pragma unroll
          for (int i = LO[ST >= 0 ? ST : 0]; i < LO[ST >= 0 ? ST + 1 : 0]; ++i)
            *reinterpret_cast<u32x4*>(oq_ptr + i * oq_step) = oq[i];
        }
      }
      if (PIPE && HAS_P && LD >= 0) {
#pragma unroll
        for (int i = LO[LD >= 0 ? LD : 0]; i < LO[LD >= 0 ? LD + 1 : 0]; ++i) pr[i] = *reinterpret_cast<const u32x4*>(pp0 + i * pstep);
      }
      const bool stamping = kDebugSwitches && a.stamps != nullptr && blockIdx.x == 0 && ti == 1;
      unsigned long long* sp = stamping ? a.stamps + ((size_t)wave * 64 + (size_t)(stamp_step & 15)) * 4 : nullptr;
      if (stamping && lane == 0) sp[0] = mg_stamp();
      const bf16_t* sB0 = srcB(cB, 0);
      const bf16_t* sB1 = srcB(cB, 1);
      const bf16_t* sA0 = srcA(cA, 0);
      const bf16_t* sA1 = srcA(cA, 1);
      const int pb1 = wrap1(posB + 1), pa1 = wrap1(posA + 1);
      const char* sa = smem + wrap1(p0 + wm) * MG_SUB;        // activation rows (m): lanes
      const char* sb = smem + wrap1(p0 + 2 + (wn >> 1)) * MG_SUB + (wn & 1) * 8192;   // weight rows (n): accumulator registers
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 fw[2], fx[4];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) fw[nb] = *reinterpret_cast<const bf16x8*>(sb + nb * 4096 + offk[kk]);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) fx[mb] = *reinterpret_cast<const bf16x8*>(sa + mb * 4096 + offk[kk]);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[nb], fx[mb], acc[nb][mb], 0, 0, 0);
        // this k block's share of the DMA: kk 0, 1 -> the two B sub-slots of step g + 1, kk 2, 3 -> the two A sub-slots of g + 2
        if (!(PIPE && LAST)) {
          if (kk == 0) issue_sub(sB0, voffB, posB);
          if (kk == 1) issue_sub(sB1, voffB, pb1);
          if (kk == 2) issue_sub(sA0, voffA, posA);
          if (kk == 3) issue_sub(sA1, voffA, pa1);
        } else {
          if (kk == 0) {
            issue_sub(sB0, voffB, posB);
            issue_sub(sB1, voffB, pb1);
          }
          if (kk == 1) {
            issue_sub(sA0, voffA, posA);
            issue_sub(sA1, voffA, pa1);
          }
        }
      }
      advance(cB);
      bump(posB);
      advance(cA);
      bump(posA);
      bump(p0);
      // step g + 1 must have landed before its first read; the only younger pieces of this wave are the A half of step g + 2
      // (4 instructions).  (Loads and stores issued before this step's DMA are older: the wait covers them.)
      if (stamping && lane == 0) sp[1] = mg_stamp();
      if (PIPE && LAST) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
      else mg_wait_vmcnt<4>();
      if (stamping && lane == 0) sp[2] = mg_stamp();
      __builtin_amdgcn_s_barrier();   // every wave's pieces of step g + 1 are in LDS; every wave is done reading step g
      asm volatile("" ::: "memory");
      if (stamping && lane == 0) sp[3] = mg_stamp();
      ++stamp_step;
    };
    typedef std::integral_constant<int, -1> No;
    if (PIPE) {   // host: nk >= 12
      k_step(std::integral_constant<int, 0>{}, No{}, std::false_type{});
      k_step(std::integral_constant<int, 1>{}, No{}, std::false_type{});
      k_step(std::integral_constant<int, 2>{}, No{}, std::false_type{});
      k_step(std::integral_constant<int, 3>{}, No{}, std::false_type{});
      k_step(std::integral_constant<int, 4>{}, No{}, std::false_type{});
      k_step(std::integral_constant<int, 5>{}, No{}, std::false_type{});
      for (int kt = 12; kt < nk; ++kt) k_step(No{}, No{}, std::false_type{});
      k_step(No{}, std::integral_constant<int, 0>{}, std::false_type{});
      k_step(No{}, std::integral_constant<int, 1>{}, std::false_type{});
      k_step(No{}, std::integral_constant<int, 2>{}, std::false_type{});
      k_step(No{}, std::integral_constant<int, 3>{}, std::false_type{});
      k_step(No{}, std::integral_constant<int, 4>{}, std::false_type{});
      k_step(No{}, std::integral_constant<int, 5>{}, std::true_type{});
    } else {
      for (int kt = 0; kt < nk - 1; ++kt) k_step(No{}, No{}, std::false_type{});
      if (HAS_P) {
        // the first of the epilogue's two dependent HBM reads, requested one K step ahead: older than the last step's LDS-DMA, so that
        // step's closing s_waitcnt vmcnt(4) covers it and it travels behind the step's MFMAs
#pragma unroll
        for (int p = 0; p < 8; ++p) pr[p] = *reinterpret_cast<const u32x4*>(pp0 + p * pstep);
      }
      k_step(No{}, No{}, std::false_type{});
    }
    {
      // ---- epilogue of tile (ctm, ctn): acc[nb][mb][e] = C[m = 128 wm + 32 mb + r][n = 64 wn + 32 nb + 8 (e>>2) + 4 h + (e&3)]
      // A lane holds pieces of 32 different rows, and row-per-lane global accesses are issue-bound (16 stores of 16 bytes per lane
      // took 4.7 us per tile and wave pair: the whole difference to the library on the plain product, and as much again for the
      // loads of P).  So every global access of the epilogue is a whole 128-byte line per 8 lanes -- one wave-instruction = 8 rows
      // x 128 B -- and the tile changes layout through a wave-private LDS image: the four sub-slots of the step just finished are
      // free until step g + 1 issues its first DMA (the barrier at the end of this block), 8 KiB = 64 rows x 128 B per wave,
      // 16-byte chunk index ^= row & 7 (both access shapes conflict-free or 2-way).  No workgroup barrier inside: a wave's LDS
      // operations execute in order.
      const int pfree = p0 >= 4 ? p0 - 4 : p0 + 6;
      char* stg = smem + wrap1(pfree + (wave >> 1)) * MG_SUB + (wave & 1) * 8192;
      const int m0 = ctm * MG_TILE + 128 * wm;
      const int n_base = ctn * MG_TILE + 64 * wn;
      const int sw = r & 7, L3 = lane >> 3, L7 = lane & 7;
      char* acc_ptr = stg + r * 128 + 8 * h;                     // + mb' * 4096 + (((4 nb + q) ^ sw) << 4): 4 columns of row 32 mb' + r
      char* row_ptr = stg + L3 * 128 + ((L7 ^ L3) << 4);          // + p * 1024: chunk L7 of row 8 p + L3
      // Every load of the epilogue is issued before its first store: vmcnt retires in order, so waiting for a load that was issued
      // behind stores waits for those stores to drain as well (the bias read piecemeal between the two halves' stores cost 60 us of
      // a 1190 us launch).
      const bool has_bias = (MODE == MG_FWD_ACT || MODE == MG_FWD_ACT_G || MODE == MG_BWD_DACT) && a.bias != nullptr;
      float bv[2][4][4];   // bias of this lane's columns 32 nb + 8 q + 4 h + i
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
          if (has_bias) t = *reinterpret_cast<const float4*>(a.bias + n_base + 32 * nb + 8 * q + 4 * h);
          bv[nb][q][0] = t.x; bv[nb][q][1] = t.y; bv[nb][q][2] = t.z; bv[nb][q][3] = t.w;
        }
      float csr[8];   // MG_BWD_DACT: sums of this lane's 8 columns (chunk L7) over the rows it stores, taken from the ROUNDED values
#pragma unroll
      for (int c = 0; c < 8; ++c) csr[c] = 0.f;
      // One pass = RP rows of the wave's 128: 64 (one 8-KiB image) for the one-output modes; 32 for the forward modes with two outputs
      // (two 4-KiB images side by side), so that act and the second output are computed once and leave together.
      constexpr bool TWO = MODE == MG_FWD_ACT || MODE == MG_FWD_ACT_G;
      constexpr int RP = TWO ? 32 : 64, NPASS = 128 / RP, MBL = RP / 32, NPIECE = RP / 8;
      const bool two = TWO && a.C2 != nullptr;
#pragma unroll
      for (int hb = 0; hb < NPASS; ++hb) {
        if (HAS_P) {
#pragma unroll
          for (int p = 0; p < 8; ++p) *reinterpret_cast<u32x4*>(row_ptr + p * 1024) = pr[PIPE ? 8 * hb + p : p];
          if (!PIPE && hb == 0) {   // the second half's block: requested now, needed after the first half's arithmetic and stores
#pragma unroll
            for (int p = 0; p < 8; ++p) pr[p] = *reinterpret_cast<const u32x4*>(pp0 + (8 + p) * pstep);
          }
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float bvq[4] = {bv[nb][q][0], bv[nb][q][1], bv[nb][q][2], bv[nb][q][3]};
#pragma unroll
            for (int mbl = 0; mbl < MBL; ++mbl) {
              const f32x16 tile = acc[nb][MBL * hb + mbl];
              char* ap = acc_ptr + mbl * 4096 + (((4 * nb + q) ^ sw) << 4);
              float y[4], y2[4];
              if (MODE == MG_FWD_ACT_G) {
#pragma unroll
                for (int i = 0; i < 4; ++i) mg_act_both<ACT>(tile[4 * q + i] + bvq[i], y[i], y2[i]);
              } else if (MODE == MG_FWD_ACT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  y2[i] = tile[4 * q + i];
                  y[i] = (dbg & 8) ? y2[i] : mg_act<ACT>(y2[i] + bvq[i]);
                }
              } else if (MODE == MG_PLAIN) {
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = tile[4 * q + i];
              } else {
                const uint2 pz = *reinterpret_cast<const uint2*>(ap);
                const float z[4] = {__uint_as_float(pz.x << 16), __uint_as_float(pz.x & 0xffff0000u), __uint_as_float(pz.y << 16),
                                    __uint_as_float(pz.y & 0xffff0000u)};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  y[i] = MODE == MG_BWD_MUL ? tile[4 * q + i] * z[i]
                                            : ((dbg & 8) ? tile[4 * q + i] + z[i] : tile[4 * q + i] * mg_act_grad<ACT>(z[i] + bvq[i]));
              }
              *reinterpret_cast<uint2*>(ap) = make_uint2(mg_pk(y[0], y[1]), mg_pk(y[2], y[3]));
              if (TWO && two) *reinterpret_cast<uint2*>(ap + 4096) = make_uint2(mg_pk(y2[0], y2[1]), mg_pk(y2[2], y2[3]));
            }
          }
        }
        bf16_t* out = a.C + (size_t)(m0 + RP * hb + L3) * a.ldc + n_base + 8 * L7;
        bf16_t* out2 = TWO && two ? a.C2 + (size_t)(m0 + RP * hb + L3) * a.ldc + n_base + 8 * L7 : nullptr;
        const long ostep = 8 * a.ldc;
#pragma unroll
        for (int p = 0; p < NPIECE; ++p) {
          const uint4 v = *reinterpret_cast<const uint4*>(row_ptr + p * 1024);
          if (PIPE) oq[NPIECE * hb + p] = u32x4{v.x, v.y, v.z, v.w};
          else if (!(dbg & 4)) *reinterpret_cast<uint4*>(out) = v;
          out += ostep;
          if (TWO && two) {
            const uint4 v2 = *reinterpret_cast<const uint4*>(row_ptr + 4096 + p * 1024);
            if (!(dbg & 4)) *reinterpret_cast<uint4*>(out2) = v2;
            out2 += ostep;
          }
          if (MODE == MG_BWD_DACT || MODE == MG_BWD_MUL) {
            csr[0] += __uint_as_float(v.x << 16); csr[1] += __uint_as_float(v.x & 0xffff0000u);
            csr[2] += __uint_as_float(v.y << 16); csr[3] += __uint_as_float(v.y & 0xffff0000u);
            csr[4] += __uint_as_float(v.z << 16); csr[5] += __uint_as_float(v.z & 0xffff0000u);
            csr[6] += __uint_as_float(v.w << 16); csr[7] += __uint_as_float(v.w & 0xffff0000u);
          }
        }
      }
      if ((MODE == MG_BWD_DACT || MODE == MG_BWD_MUL) && a.part != nullptr) {
        // lanes L7 + 8 k (k = lane >> 3) hold partial sums of the same 8 columns: transpose-reduce over k (7 exchanges), column
        // index c ends on the lane whose k has bit pattern c
#pragma unroll
        for (int s2 = 4; s2 >= 1; s2 >>= 1) {
          const bool up = (L3 & s2) != 0;
#pragma unroll
          for (int k = 0; k < s2; ++k) {
            const float send = up ? csr[k] : csr[k + s2];
            const float keep = up ? csr[k + s2] : csr[k];
            csr[k] = keep + __shfl_xor(send, 8 * s2);
          }
        }
        a.part[(size_t)(2 * ctm + wm) * a.N + n_base + 8 * L7 + L3] = csr[0];
      }
      if (kDebugSwitches && a.stamps != nullptr && blockIdx.x == 0 && ti == 1 && lane == 0) a.stamps[((size_t)wave * 64 + 20) * 4] = mg_stamp();
      if (PIPE) oq_ptr = (dbg & 4) ? nullptr : a.C + (size_t)(m0 + L3) * a.ldc + n_base + 8 * L7;
      // every wave is done with its staging image before any wave lets step g + 1's DMA into these sub-slots
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kDebugSwitches && a.stamps != nullptr && blockIdx.x == 0 && ti == 1 && lane == 0) a.stamps[((size_t)wave * 64 + 21) * 4] = mg_stamp();
    }
    next_tile(ctm, ctn);
  }
  if (PIPE && oq_ptr != nullptr) {   // the last tile's pieces
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x4*>(oq_ptr + i * oq_step) = oq[i];
  }
  mg_wait_vmcnt<0>();   // the ring schedule's last (unused) pieces must not land in LDS after the workgroup has gone
}

}  // namespace mmk

using namespace mmk;

extern "C" {

// 1 when the MLP GEMMs serve the shape; otherwise the caller keeps library GEMM + bias_act kernel
int mmk_mlp_gemm_supported(int64_t M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc) {
  return M >= MG_TILE && M % MG_TILE == 0 && N >= MG_TILE && N % MG_TILE == 0 && K >= 2 * MG_BK && K % MG_BK == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
         ldc % 8 == 0 && M < (1ll << 31) - 512 && (int64_t)127 * std::max(lda, ldb) * 2 + 128 < (1ll << 32);
}

int mmk_mlp_gemm_part_rows(int64_t M) { return (int)(2 * ((M + MG_TILE - 1) / MG_TILE)); }

static int mlp_gemm_launch(int mode, int act, MlpGemmArgs& a, hipStream_t st) {
  a.tiles_m = cdiv(a.M, MG_TILE);
  a.tiles_n = cdiv(a.N, MG_TILE);
  a.dbg = MMK_DBG_ENV("MMK_MLP_GEMM_DBG") ? atoi(MMK_DBG_ENV("MMK_MLP_GEMM_DBG")) : 0;
  a.stamps = MMK_DBG_ENV("MMK_MLP_GEMM_STAMPS") ? reinterpret_cast<unsigned long long*>(strtoull(MMK_DBG_ENV("MMK_MLP_GEMM_STAMPS"), nullptr, 0)) : nullptr;
  const int total = a.tiles_m * a.tiles_n;
  // PIPE for the backward only: on the plain product it measured 894 vs 899 us (M = 201,728), i.e. nothing -- the C stores cost
  // the launch ~130 us whenever they are issued (MMK_MLP_GEMM_DBG=4 removes them: 799 us) -- while with G to fetch it is
  // 1130 vs 1189 us
  const bool pipe = (mode == MG_BWD_MUL || (mode == MG_PLAIN && (a.dbg & 1024))) && a.K / MG_BK >= 12 && !(a.dbg & 512);
  const void* kern = nullptr;
#define MG_PICK(MODE_, ACT_, PIPE_) \
  if (mode == MODE_ && act == ACT_ && pipe == PIPE_) kern = reinterpret_cast<const void*>(mlp_gemm_kernel<MODE_, ACT_, PIPE_>);
  MG_PICK(MG_PLAIN, 0, false) MG_PICK(MG_PLAIN, 0, true)
  MG_PICK(MG_FWD_ACT, MG_ACT_QUICK_GELU, false) MG_PICK(MG_FWD_ACT, MG_ACT_GELU, false)
  MG_PICK(MG_BWD_DACT, MG_ACT_QUICK_GELU, false) MG_PICK(MG_BWD_DACT, MG_ACT_GELU, false)
  MG_PICK(MG_FWD_ACT_G, MG_ACT_QUICK_GELU, false) MG_PICK(MG_FWD_ACT_G, MG_ACT_GELU, false)
  MG_PICK(MG_BWD_MUL, 0, false) MG_PICK(MG_BWD_MUL, 0, true)
#undef MG_PICK
  MMK_REQUIRE(kern != nullptr, "mlp_gemm: unknown (mode, activation)");
  KernelSetup ks;   // > 64 KiB LDS opt-in and the CU count, per device and kernel
  if (int rc = kernel_setup(kern, 512, MG_LDS, &ks)) return rc;
  const int grid = std::min(std::max(8, ks.cus / 8 * 8), round_up(total, 8));
  {
    ProfEvents pe(MMK_K_MLP_GEMM);
    void* params[] = {&a};
    MMK_HIP(hipExtLaunchKernel(kern, dim3(grid), dim3(512), params, MG_LDS, st, pe.start, pe.stop, 0));
  }
  MMK_LAUNCH_CHECK();
  return 0;
}

// C = A B^T (bf16 out): the main loop alone, for A/B timings against the library (tools/bench_mlp_fusion.py) and parity tests
int mmk_mlp_gemm_plain(const void* A, const void* B, void* C, int64_t M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, void* stream) {
  MMK_REQUIRE(A && B && C, "null pointer");
  MMK_REQUIRE(mmk_mlp_gemm_supported(M, N, K, lda, ldb, ldc), "mlp_gemm: unsupported shape (need M % 256 == 0, N % 256 == 0, K % 64 == 0, strides % 8 == 0)");
  MlpGemmArgs a = {};
  a.A = static_cast<const bf16_t*>(A); a.B = static_cast<const bf16_t*>(B); a.C = static_cast<bf16_t*>(C);
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldp = 0; a.M = (int)M; a.N = N; a.K = K;
  return mlp_gemm_launch(MG_PLAIN, 0, a, static_cast<hipStream_t>(stream));
}

// H = act(X W^T + bias) and (pre nullable) the bias-free pre-activation X W^T, both bf16 with row stride ldc
int mmk_mlp_gemm_fwd_act(const void* X, const void* W, const float* bias, void* H, void* pre, int64_t M, int N, int K, int64_t ldx, int64_t ldw,
                         int64_t ldc, int act, void* stream) {
  MMK_REQUIRE(X && W && H, "null pointer");
  MMK_REQUIRE(act == MG_ACT_QUICK_GELU || act == MG_ACT_GELU, "mlp_gemm: act must be 0 (quick_gelu) or 1 (gelu)");
  MMK_REQUIRE(mmk_mlp_gemm_supported(M, N, K, ldx, ldw, ldc), "mlp_gemm: unsupported shape (need M % 256 == 0, N % 256 == 0, K % 64 == 0, strides % 8 == 0)");
  MlpGemmArgs a = {};
  a.A = static_cast<const bf16_t*>(X); a.B = static_cast<const bf16_t*>(W); a.C = static_cast<bf16_t*>(H); a.C2 = static_cast<bf16_t*>(pre);
  a.bias = bias;
  a.lda = ldx; a.ldb = ldw; a.ldc = ldc; a.ldp = 0; a.M = (int)M; a.N = N; a.K = K;
  return mlp_gemm_launch(MG_FWD_ACT, act, a, static_cast<hipStream_t>(stream));
}

// dPre = (dY Wt^T) * act'(pre + bias): dY [M, K], Wt [N, K] (= fc2.weight^T, K-contiguous), pre [M, N] row stride ldp.
// part (nullable): f32[mmk_mlp_gemm_part_rows(M)][N], row i = column sums of dPre over output rows [128 i, 128 i + 128).
int mmk_mlp_gemm_bwd_dact(const void* dY, const void* Wt, const void* pre, const float* bias, void* dPre, float* part, int64_t M, int N, int K,
                          int64_t ldy, int64_t ldw, int64_t ldp, int64_t ldc, int act, void* stream) {
  MMK_REQUIRE(dY && Wt && pre && dPre, "null pointer");
  MMK_REQUIRE(act == MG_ACT_QUICK_GELU || act == MG_ACT_GELU, "mlp_gemm: act must be 0 (quick_gelu) or 1 (gelu)");
  MMK_REQUIRE(mmk_mlp_gemm_supported(M, N, K, ldy, ldw, ldc) && ldp % 8 == 0,
              "mlp_gemm: unsupported shape (need M % 256 == 0, N % 256 == 0, K % 64 == 0, strides % 8 == 0)");
  MlpGemmArgs a = {};
  a.A = static_cast<const bf16_t*>(dY); a.B = static_cast<const bf16_t*>(Wt); a.C = static_cast<bf16_t*>(dPre);
  a.P = static_cast<const bf16_t*>(pre); a.bias = bias; a.part = part;
  a.lda = ldy; a.ldb = ldw; a.ldc = ldc; a.ldp = ldp; a.M = (int)M; a.N = N; a.K = K;
  return mlp_gemm_launch(MG_BWD_DACT, act, a, static_cast<hipStream_t>(stream));
}

// H = act(X W^T + bias), G = act'(X W^T + bias), both bf16 with row stride ldc: the forward of the pair the product runs
int mmk_mlp_gemm_fwd_act_grad(const void* X, const void* W, const float* bias, void* H, void* G, int64_t M, int N, int K, int64_t ldx, int64_t ldw,
                              int64_t ldc, int act, void* stream) {
  MMK_REQUIRE(X && W && H && G, "null pointer");
  MMK_REQUIRE(act == MG_ACT_QUICK_GELU || act == MG_ACT_GELU, "mlp_gemm: act must be 0 (quick_gelu) or 1 (gelu)");
  MMK_REQUIRE(mmk_mlp_gemm_supported(M, N, K, ldx, ldw, ldc), "mlp_gemm: unsupported shape (need M % 256 == 0, N % 256 == 0, K % 64 == 0, strides % 8 == 0)");
  MlpGemmArgs a = {};
  a.A = static_cast<const bf16_t*>(X); a.B = static_cast<const bf16_t*>(W); a.C = static_cast<bf16_t*>(H); a.C2 = static_cast<bf16_t*>(G);
  a.bias = bias;
  a.lda = ldx; a.ldb = ldw; a.ldc = ldc; a.ldp = 0; a.M = (int)M; a.N = N; a.K = K;
  return mlp_gemm_launch(MG_FWD_ACT_G, act, a, static_cast<hipStream_t>(stream));
}

// dPre = (dY Wt^T) * G elementwise (G [M, N] bf16, row stride ldg, as mmk_mlp_gemm_fwd_act_grad left it); part as in mmk_mlp_gemm_bwd_dact
int mmk_mlp_gemm_bwd_mul(const void* dY, const void* Wt, const void* G, void* dPre, float* part, int64_t M, int N, int K, int64_t ldy,
                         int64_t ldw, int64_t ldg, int64_t ldc, void* stream) {
  MMK_REQUIRE(dY && Wt && G && dPre, "null pointer");
  MMK_REQUIRE(mmk_mlp_gemm_supported(M, N, K, ldy, ldw, ldc) && ldg % 8 == 0,
              "mlp_gemm: unsupported shape (need M % 256 == 0, N % 256 == 0, K % 64 == 0, strides % 8 == 0)");
  MlpGemmArgs a = {};
  a.A = static_cast<const bf16_t*>(dY); a.B = static_cast<const bf16_t*>(Wt); a.C = static_cast<bf16_t*>(dPre);
  a.P = static_cast<const bf16_t*>(G); a.part = part;
  a.lda = ldy; a.ldb = ldw; a.ldc = ldc; a.ldp = ldg; a.M = (int)M; a.N = N; a.K = K;
  return mlp_gemm_launch(MG_BWD_MUL, 0, a, static_cast<hipStream_t>(stream));
}

}
