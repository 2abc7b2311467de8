// SURVEY 8(f3) -- eval-side similarity + top-k membership: RetrievalRecallAtK (mmlearn/modules/metrics/
// retrieval_recall.py:171-289) builds, per batch of queries, the [b, M] cosine-score matrix against the whole database,
// runs torch.topk on it and looks the single positive of every query up among the k winners -- on the CPU, in a thread
// pool.  "Is the positive among the top k" only needs the RANK of the positive, i.e. how many database rows score higher;
// that is the similarity GEMM with a counting epilogue, and nothing N x M is ever stored:
//   pass 0: the tile that holds column pos[i] of row i records t_i = s(i, pos_i);
//   pass 1: every tile counts, per row, the columns with s > t_i (ties: the lower database index wins, the order of a
//           stable descending sort; torch.topk leaves tie order unspecified) and adds the count to rank[i].
// Both passes run the SAME tile arithmetic (v_mfma_f32_32x32x2_f32 over the same k order), so a score and its own
// threshold compare bit-exactly -- duplicates of the positive in the database tie instead of flipping on rounding.
// f32 in, f32 MFMA: the metric is an exact count, bf16 scores would reorder near neighbours.
#include "common.h"

namespace mmk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int RK_TILE = 128;  // queries x database rows per workgroup: 4 waves as 2 x 2, 64 x 64 (= 2 x 2 MFMA tiles) each
constexpr int RK_KC = 32;     // floats of the contraction per LDS stage

// MODE 0 runs only where a positive lives (a workgroup whose 128 x 128 tile holds none exits before loading anything);
// MODE 1 runs everywhere.  Arithmetic intensity is what the tile size buys: 64 x 64 tiles re-read the operands from
// L2 / Infinity Cache 400x at n = 25,000 and ran at 36 TFLOP/s on that traffic alone.
template <int MODE>
__global__ __launch_bounds__(256) void recall_rank_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const int64_t* __restrict__ pos, float* __restrict__ tpos,
                                                          int32_t* __restrict__ rank, int N, int M, int D) {
  __shared__ float xs[RK_TILE][RK_KC + 1], ys[RK_TILE][RK_KC + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int q0 = blockIdx.y * RK_TILE, d0 = blockIdx.x * RK_TILE;
  if (MODE == 0) {
    bool mine = false;
    if (tid < RK_TILE && q0 + tid < N) {
      const long p = pos[q0 + tid];
      mine = p >= d0 && p < d0 + RK_TILE;
    }
    if (!__syncthreads_or(mine)) return;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // staging: thread t moves 16 floats of row t/2 per operand and chunk (two threads per row, 4 float4 each)
  const int lrow = tid >> 1, lseg = (tid & 1) * 16;
  const bool vec = (D & 3) == 0;
  float4 ra[4], rb[4];
  auto fetch = [&](int k0) {  // the next chunk travels in registers while the MFMAs of the current one run
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int k = k0 + lseg + 4 * v;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (vec) {
        if (q0 + lrow < N && k < D) a = *reinterpret_cast<const float4*>(x + (long)(q0 + lrow) * D + k);
        if (d0 + lrow < M && k < D) b = *reinterpret_cast<const float4*>(y + (long)(d0 + lrow) * D + k);
      } else {
        float av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (q0 + lrow < N && k + j < D) av[j] = x[(long)(q0 + lrow) * D + k + j];
          if (d0 + lrow < M && k + j < D) bv[j] = y[(long)(d0 + lrow) * D + k + j];
        }
        a = make_float4(av[0], av[1], av[2], av[3]);
        b = make_float4(bv[0], bv[1], bv[2], bv[3]);
      }
      ra[v] = a;
      rb[v] = b;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < D; k0 += RK_KC) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float* xd = &xs[lrow][lseg + 4 * v];
      float* yd = &ys[lrow][lseg + 4 * v];
      xd[0] = ra[v].x; xd[1] = ra[v].y; xd[2] = ra[v].z; xd[3] = ra[v].w;
      yd[0] = rb[v].x; yd[1] = rb[v].y; yd[2] = rb[v].z; yd[3] = rb[v].w;
    }
    __syncthreads();
    if (k0 + RK_KC < D) fetch(k0 + RK_KC);
#pragma unroll
    for (int kk = 0; kk < RK_KC / 2; ++kk) {  // lane (r, h): A[row r][k = 2kk + h], B[k = 2kk + h][col r]
      float af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = xs[64 * wm + 32 * i + r][2 * kk + h];
        bf[i] = ys[64 * wn + 32 * i + r][2 * kk + h];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // acc[i][j][e] = s(row, col): row = q0 + 64 wm + 32 i + (e&3) + 8(e>>2) + 4h, col = d0 + 64 wn + 32 j + (lane & 31)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = q0 + 64 * wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
      const long p = row < N ? pos[row] : -1;
      const float t = (MODE == 1 && row < N) ? tpos[row] : 0.f;
      int c = 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = d0 + 64 * wn + 32 * j + r;
        const bool in = row < N && col < M;
        const float sv = acc[i][j][e];
        if (MODE == 0) {
          if (in && col == p) tpos[row] = sv;
        } else {
          const bool beats = in && col != p && (sv > t || (sv == t && col < p));
          const unsigned long long m = __ballot(beats);
          c += h == 0 ? __popcll(m & 0xFFFFFFFFull) : __popcll(m >> 32);
        }
      }
      if (MODE == 1 && r == 0 && row < N && c) atomicAdd(rank + row, c);
    }
}

}  // namespace mmk

using namespace mmk;

extern "C" int mmk_recall_ranks(const float* x, const float* y, const int64_t* pos, float* tpos_ws, int32_t* rank, int n, int m, int d,
                                void* stream) {
  // x: [n, d], y: [m, d] f32 row-major (already L2-normalised by the caller); pos: int64[n] in [0, m); tpos_ws: f32[n];
  // rank (out): int32[n] = number of database rows ranked before the positive
  MMK_REQUIRE(x && y && pos && tpos_ws && rank && n > 0 && m > 0 && d > 0, "bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  MMK_HIP(hipMemsetAsync(rank, 0, sizeof(int32_t) * n, st));
  const dim3 grid(cdiv(m, RK_TILE), cdiv(n, RK_TILE));
  hipLaunchKernelGGL((recall_rank_kernel<0>), grid, dim3(256), 0, st, x, y, pos, tpos_ws, rank, n, m, d);
  MMK_LAUNCH_CHECK();
  {
    ProfScope ps(MMK_K_RECALL, st);
    hipLaunchKernelGGL((recall_rank_kernel<1>), grid, dim3(256), 0, st, x, y, pos, tpos_ws, rank, n, m, d);
  }
  MMK_LAUNCH_CHECK();
  return 0;
}
