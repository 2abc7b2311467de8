// forward / input-gradient GEMM of the encoders' Linear layers for gfx950 (MI355X), second design,
//
//     C[M, N] = A[M, K] . B[N, K]^T  (+ bias[N]) (-> activation)        bf16 operands, f32 accumulation
//
// (nn.Linear's  y = x W^T + b  with A = x, B = W, and its  dX = dY W  with A = dY, B = W^T; mmlearn/modules/encoders/clip.py:29-470,
// text.py:20-178, modules/layers/{attention,mlp}.py).  csrc/gemm.hip runs its eight waves as two groups half a phase apart,
// four barriers per K step; measured with NO memory traffic at all that structure tops out at 1.36 PFLOP/s.  Two probes
// (tools/probes/gemm4_inner.hip: operands re-read from a fixed LDS image, one barrier per K step, everything else left to the
// compiler's scheduling) say the phases are the limit, not the hardware:
//     four waves (one per SIMD), 128 x 128 per wave, 256 accumulator registers in the AGPR half of the file: 2.10 PFLOP/s
//     eight waves (two per SIMD), 128 x 64 per wave, free running:                                          2.30 PFLOP/s
// This kernel keeps gemm.hip's data movement (persistent workgroups, ring of ten 16-KiB LDS-DMA sub-slots, counted vmcnt,
// XCD-aware tile order, register-direct 16-byte C stores) and replaces the inner loop by the second probe's: eight waves as
// 2 (m) x 4 (n), ONE barrier per K step, fragment reads and MFMAs scheduled by the compiler, the step's eight LDS-DMA
// instructions per wave spread over its four k blocks.
//
// RESULT (round 2, [201728 x K] . [K x N], K, N in {768, 2304, 3072}): 0.83-1.05 PFLOP/s = the speed of gemm.hip and 0.85-0.97 x
// hipBLASLt, so it is NOT dispatched either; the probes' 2.3 PFLOP/s disappears as soon as the operands move.  Ablations
// (MMK_GEMM4_DBG): no C stores 1.01-1.13; every DMA re-reading one cached 64 KiB 0.95-1.20; both 1.13-1.25.  The same ceiling
// with every other loader tried on this skeleton (tools/probes/*.txt): four waves + ring 0.85-1.10 (hot, no stores 1.0-1.18),
// four waves + global_load -> registers -> ds_write_b128 one step ahead 0.63-0.83, eight waves + the same register staging
// 0.76-0.97 (no stores 1.0-1.13), eight waves + ring with a K step of 32 and FOUR steps in flight 0.75-0.99 (no stores 1.0-1.09):
// neither the LDS-DMA issue cost nor the prefetch distance is the limit.  The probe says what is: with eight LDS-DMA
// instructions per step and wave that nobody ever waits for, the MFMA loop still runs at 2.13 PFLOP/s while they re-read one
// hot 8 KiB, and at 1.49-1.55 PFLOP/s when every workgroup streams its own data (64 KiB re-read every step = L2-resident, or
// 256 KiB = Infinity-Cache-resident, with or without a counted vmcnt; twice the bytes per step: 1.22 = 18.6 TB/s, the saturated
// fill rate): feeding LDS from beyond the L1 at the 11-12 TB/s a 256 x 256 tile needs (one byte per 128 flop) costs a third of
// the MFMA rate, so ~1.5
// PFLOP/s is the ceiling of ANY LDS-staged 256 x 256 bf16 GEMM here, and barriers / waits / the C stores take it to the
// 1.0-1.25 that this kernel, gemm.hip and hipBLASLt all reach.  Requires M % 256 == N % 256 == 0.
#include <hip/hip_ext.h>
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace mmk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));

constexpr int G4_TILE = 256;          // output tile edge
constexpr int G4_BK = 64;             // K per step
constexpr int G4_SUB = 128 * 128;     // bytes of one sub-slot: 128 rows x 64 bf16
constexpr int G4_RING = 10;           // sub-slots in the ring
constexpr int G4_LDS = G4_RING * G4_SUB;

enum { G4_ACT_NONE = 0, G4_ACT_QUICK_GELU = 1, G4_ACT_GELU = 2 };

struct Lin4Args {
  const bf16_t* A;    // [M, K] row stride lda
  const bf16_t* B;    // [N, K] row stride ldb
  void* C;            // [M, N] row stride ldc (bf16 or f32)
  void* C2;           // optional second output: the pre-activation values (same dtype / stride as C), or null
  const float* bias;  // [N] or null
  long lda, ldb, ldc;
  int M, N, K;
  int tiles_m, tiles_n;
  int dbg;   // timing ablations only (MMK_GEMM4_DBG): 4 = no C stores, 64 = every DMA re-reads the same 64 KiB
};

__device__ __forceinline__ void g4_dma16(const void* sbase, uint32_t voff, uint32_t lds_addr) {
  const uint64_t pb = reinterpret_cast<uint64_t>(sbase);
  const uint64_t ps = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pb >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pb);
  lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  // no "memory" clobber: the DMA lands in ring slots nobody reads during this step (the barriers order it), and the compiler
  // must stay free to move this step's fragment reads across it
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(reinterpret_cast<const void*>(ps)), "s"(lds_addr));
}

template <int N>
__device__ __forceinline__ void g4_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ float g4_act(float x, int act) {
  if (act == G4_ACT_QUICK_GELU) return x / (1.f + __expf(-1.702f * x));
  if (act == G4_ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
  return x;
}

// pack two f32 into one dword of two bf16 (RNE, NaN-preserving: v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t g4_pk(float lo, float hi) {
  typedef bf16_t bf2 __attribute__((ext_vector_type(2)));
  bf2 v;
  v[0] = (bf16_t)lo;
  v[1] = (bf16_t)hi;
  return __builtin_bit_cast(uint32_t, v);
}

template <int OUT_F32, int ACT, int HAS_C2>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void lin4_gemm_kernel(const Lin4Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // wave tile: output rows 128 wm .., output columns 64 wn ..
  const int r = lane & 31, h = lane >> 5;
  const uint32_t ring = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)smem);

  // ---- this workgroup's tiles: XCD x = blockIdx % 8 takes tiles [(i * 8 + x) * per_xcd, + per_xcd), i = 0, 1, ...
  const int per_xcd = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int total = a.tiles_m * a.tiles_n;
  const int nk = a.K / G4_BK;
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
    if (pos >= G4_RING) pos -= G4_RING;
  };
  auto wrap1 = [&](int pos) { return pos >= G4_RING ? pos - G4_RING : pos; };
  auto issue_sub = [&](const bf16_t* src, const uint32_t (&voff)[2], int pos) {   // src: first row of the 128-row half, k position applied
    const uint32_t dst = wave_lds + (uint32_t)pos * G4_SUB;
#pragma unroll
    for (int u = 0; u < 2; ++u) g4_dma16(src, voff[u], dst + (uint32_t)u * 1024u);
  };
  const bool hot = (a.dbg & 64) != 0;   // ablation: every DMA re-reads the same (cache-resident) 64 KiB
  auto srcA = [&](const Cur& c, int half) { return a.A + (hot ? 0 : ((long)c.tm * G4_TILE + 128 * half) * a.lda + (long)c.kt * G4_BK); };
  auto srcB = [&](const Cur& c, int half) { return a.B + (hot ? 0 : ((long)c.tn * G4_TILE + 128 * half) * a.ldb + (long)c.kt * G4_BK); };

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
  g4_wait_vmcnt<4>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
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
    for (int kt = 0; kt < nk; ++kt) {
      const bf16_t* sB0 = srcB(cB, 0);
      const bf16_t* sB1 = srcB(cB, 1);
      const bf16_t* sA0 = srcA(cA, 0);
      const bf16_t* sA1 = srcA(cA, 1);
      const int pb1 = wrap1(posB + 1), pa1 = wrap1(posA + 1);
      const char* sa = smem + wrap1(p0 + wm) * G4_SUB;        // activation rows (m): lanes
      const char* sb = smem + wrap1(p0 + 2 + (wn >> 1)) * G4_SUB + (wn & 1) * 8192;   // weight rows (n): accumulator registers
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
        if (kk == 0) issue_sub(sB0, voffB, posB);
        if (kk == 1) issue_sub(sB1, voffB, pb1);
        if (kk == 2) issue_sub(sA0, voffA, posA);
        if (kk == 3) issue_sub(sA1, voffA, pa1);
      }
      advance(cB);
      bump(posB);
      advance(cA);
      bump(posA);
      bump(p0);
      // step g + 1 must have landed before its first read; the only younger pieces of this wave are the A half of step g + 2
      // (4 instructions).  (Stores of an epilogue are older than this step's DMA: they have been given a whole step.)
      g4_wait_vmcnt<4>();
      __builtin_amdgcn_s_barrier();   // every wave's pieces of step g + 1 are in LDS; every wave is done reading step g
      asm volatile("" ::: "memory");
    }
    {
      // ---- epilogue of tile (ctm, ctn): acc[nb][mb][e] = C[m = 128 wm + 32 mb + r][n = 64 wn + 32 nb + (e&3) + 8 (e>>2) + 4 h]
      const int m_base = ctm * G4_TILE + 128 * wm + r;
      const int n_base = ctn * G4_TILE + 64 * wn;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        float bv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n_base + 32 * nb + 8 * q + 4 * h;
          float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
          if (a.bias != nullptr) t = *reinterpret_cast<const float4*>(a.bias + n);   // N is a multiple of the tile: no edge
          bv[4 * q] = t.x; bv[4 * q + 1] = t.y; bv[4 * q + 2] = t.z; bv[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          const int m = m_base + 32 * mb;
          const f32x16 tile = acc[nb][mb];
          float v[16], pre[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            pre[e] = tile[e] + bv[e];
            v[e] = g4_act(pre[e], ACT);
          }
          if (OUT_F32) {
            // 4 consecutive columns per register group: one 16-byte store each
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int n = n_base + 32 * nb + 8 * q + 4 * h;
              if (!(a.dbg & 4)) {
                *reinterpret_cast<float4*>(static_cast<float*>(a.C) + (size_t)m * a.ldc + n) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                if (HAS_C2)
                  *reinterpret_cast<float4*>(static_cast<float*>(a.C2) + (size_t)m * a.ldc + n) = make_float4(pre[4 * q], pre[4 * q + 1], pre[4 * q + 2], pre[4 * q + 3]);
              }
            }
          } else {
            // bf16: register group q of a lane holds columns 8q + 4h + 0..3 as two packed dwords.  For the group pair
            // (q0 = 2j, q1 = 2j + 1) one v_permlane32_swap per dword gives the lower lane (h = 0) columns 16j + 0..7 and the
            // upper lane (h = 1) columns 16j + 8..15 of the SAME row: 16 contiguous bytes per lane.
#pragma unroll
            for (int pass = 0; pass < (HAS_C2 ? 2 : 1); ++pass) {
              const float* src = pass ? pre : v;
              bf16_t* out = static_cast<bf16_t*>(pass ? a.C2 : a.C);
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                uint32_t d0[2], d1[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                  d0[i] = g4_pk(src[8 * j + 2 * i], src[8 * j + 2 * i + 1]);          // group q0: columns 16j + 4h + 2i, +1
                  d1[i] = g4_pk(src[8 * j + 4 + 2 * i], src[8 * j + 4 + 2 * i + 1]);  // group q1: columns 16j + 8 + 4h + 2i, +1
                  const auto sw = __builtin_amdgcn_permlane32_swap(d0[i], d1[i], false, false);   // d0.upper_half <-> d1.lower_half
                  d0[i] = sw[0];
                  d1[i] = sw[1];
                }
                const int n = n_base + 32 * nb + 16 * j + 8 * h;
                if (!(a.dbg & 4)) *reinterpret_cast<uint4*>(out + (size_t)m * a.ldc + n) = make_uint4(d0[0], d0[1], d1[0], d1[1]);
              }
            }
          }
        }
      }
    }
    next_tile(ctm, ctn);
  }
  g4_wait_vmcnt<0>();   // the ring schedule's last (unused) pieces must not land in LDS after the workgroup has gone
}

}  // namespace mmk

using namespace mmk;

extern "C" {

// 1 when mmk_gemm4_nt serves the shape; otherwise the caller keeps its library GEMM
int mmk_gemm4_nt_supported(int64_t M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc) {
  return M >= G4_TILE && M % G4_TILE == 0 && N >= G4_TILE && N % G4_TILE == 0 && K >= G4_BK && K % G4_BK == 0 && lda % 8 == 0 && ldb % 8 == 0 &&
         ldc % 8 == 0 && M < (1ll << 31) - 512 && (int64_t)127 * std::max(lda, ldb) * 2 + 128 < (1ll << 32);
}

int mmk_gemm4_nt(const void* A, const void* B, void* C, void* C2, const float* bias, int64_t M, int N, int K, int64_t lda, int64_t ldb,
                 int64_t ldc, int out_dtype, int act, void* stream) {
  MMK_REQUIRE(A && B && C, "null pointer");
  MMK_REQUIRE(mmk_gemm4_nt_supported(M, N, K, lda, ldb, ldc), "gemm4_nt: unsupported shape (need M % 256 == 0, N % 256 == 0, K % 64 == 0, strides % 8 == 0)");
  MMK_REQUIRE(out_dtype == MMK_BF16 || out_dtype == MMK_F32, "gemm4_nt: output must be bf16 or f32");
  MMK_REQUIRE(act >= G4_ACT_NONE && act <= G4_ACT_GELU, "gemm4_nt: unknown activation");
  MMK_REQUIRE(C2 == nullptr || act != G4_ACT_NONE, "gemm4_nt: a pre-activation output needs an activation");
  Lin4Args a;
  a.A = static_cast<const bf16_t*>(A); a.B = static_cast<const bf16_t*>(B); a.C = C; a.C2 = C2; a.bias = bias;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = (int)M; a.N = N; a.K = K;
  a.tiles_m = cdiv((int)M, G4_TILE); a.tiles_n = cdiv(N, G4_TILE);
  a.dbg = getenv("MMK_GEMM4_DBG") ? atoi(getenv("MMK_GEMM4_DBG")) : 0;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    MMK_HIP(hipGetDevice(&dev));
    MMK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    n_cu = std::max(8, n_cu / 8 * 8);
  }
  const int total = a.tiles_m * a.tiles_n;
  const int grid = std::min(n_cu, round_up(total, 8));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const void* kern = nullptr;
#define G4_PICK(F32_, ACT_, TWO_)                                                              \
  if ((out_dtype == MMK_F32) == (F32_ == 1) && act == ACT_ && (TWO_ == 1) == (a.C2 != nullptr)) \
    kern = reinterpret_cast<const void*>(lin4_gemm_kernel<F32_, ACT_, TWO_>);
  G4_PICK(0, G4_ACT_NONE, 0) G4_PICK(1, G4_ACT_NONE, 0)
  G4_PICK(0, G4_ACT_QUICK_GELU, 0) G4_PICK(0, G4_ACT_QUICK_GELU, 1)
  G4_PICK(0, G4_ACT_GELU, 0) G4_PICK(0, G4_ACT_GELU, 1)
#undef G4_PICK
  MMK_REQUIRE(kern != nullptr, "gemm4_nt: this (dtype, activation, second output) combination is not built");
  MMK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS));
  {
    ProfEvents pe(MMK_K_WGRAD);
    void* params[] = {&a};
    MMK_HIP(hipExtLaunchKernel(kern, dim3(grid), dim3(512), params, G4_LDS, st, pe.start, pe.stop, 0));
  }
  MMK_LAUNCH_CHECK();
  return 0;
}
}
