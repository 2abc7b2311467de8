// Device-side example-id matching for gfx950.
//
// Replaces find_matching_indices (mmlearn/datasets/core/example.py:160-166): the reference
// broadcasts [N,1,2]==[1,M,2] into an N*M*2 bool tensor, reduces, and calls torch.where
// (nonzero + host sync).  Here the N x M compare is a 2-D grid (256 rows of a  x  256-row chunk
// of b held in LDS and broadcast to all lanes): count pass -> single-block exclusive scan over the
// (row, chunk) counts -> fill pass.  Output order is row-major (i ascending, then j ascending),
// exactly torch.where's, and independent of scheduling (integer atomics only feed flags).
#include <algorithm>

#include "common.h"

namespace mmk {

constexpr int MATCH_CHUNK = 128;  // id pairs of b per block (2 KiB of LDS): a 256 x 128 compare tile per block

// MODE 0: cnt[chunk*n_a + i] = #{j in chunk : a[i] == b[j]}, cnt_b[j] += 1 per match.
// MODE 1: write the pairs of (i, chunk) at offs[i*n_chunks + chunk]; maintain the status flags.
template <int MODE>
__global__ __launch_bounds__(256) void match_kernel(const longlong2* __restrict__ a, int n_a, const longlong2* __restrict__ b,
                                                    int n_b, int n_chunks, int32_t* __restrict__ cnt,
                                                    int32_t* __restrict__ cnt_b, int32_t* __restrict__ idx_a,
                                                    int32_t* __restrict__ idx_b, int capacity, int32_t* __restrict__ status) {
  __shared__ longlong2 sb[MATCH_CHUNK];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int chunk = blockIdx.y;
  const int j0 = chunk * MATCH_CHUNK;
  const int len = min(MATCH_CHUNK, n_b - j0);
  for (int t = threadIdx.x; t < len; t += 256) sb[t] = b[j0 + t];
  __syncthreads();
  if (i >= n_a) return;
  const longlong2 mine = a[i];
  if (MODE == 0) {
    // compare 8 LDS entries per step into a bit mask (loads pipeline, no side effects); matches are rare, so the
    // atomics that feed the "column repeats" flag sit behind an almost-never-taken branch
    int c = 0;
    for (int t0 = 0; t0 < len; t0 += 8) {
      unsigned mask = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const longlong2 o = sb[min(t0 + e, MATCH_CHUNK - 1)];
        mask |= (unsigned)(t0 + e < len && o.x == mine.x && o.y == mine.y) << e;
      }
      c += __popc(mask);
      while (mask) {
        const int e = __ffs(mask) - 1;
        mask &= mask - 1;
        atomicAdd(&cnt_b[j0 + t0 + e], 1);
      }
    }
    cnt[(size_t)chunk * n_a + i] = c;  // [chunk][row]: coalesced for the scan and the fill pass
  } else {
    // row_off[i] = pairs before row i; the chunks of a row are laid out in chunk order behind it
    const int32_t* row_off = cnt + (size_t)n_a * n_chunks;
    int pos = row_off[i];
    for (int c = 0; c < chunk; ++c) pos += cnt[(size_t)c * n_a + i];
    if (chunk == 0 && row_off[i + 1] - pos > 1) status[2] = 1;  // row i is in several pairs
    if (cnt[(size_t)chunk * n_a + i] == 0) return;
    bool off_diag = false, col_rep = false;
    for (int t0 = 0; t0 < len; t0 += 8) {
      unsigned mask = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const longlong2 o = sb[min(t0 + e, MATCH_CHUNK - 1)];
        mask |= (unsigned)(t0 + e < len && o.x == mine.x && o.y == mine.y) << e;
      }
      while (mask) {  // ascending j: row-major order
        const int e = __ffs(mask) - 1;
        mask &= mask - 1;
        const int j = j0 + t0 + e;
        if (pos < capacity) {
          idx_a[pos] = i;
          idx_b[pos] = j;
        }
        if (pos != i || j != i) off_diag = true;
        if (cnt_b[j] > 1) col_rep = true;
        ++pos;
      }
    }
    if (off_diag) status[1] = 0;
    if (col_rep) status[3] = 1;
  }
}

// row_off[0..n] = exclusive scan of the per-row totals (sum over chunks); status = {total, identity candidate, 0, 0}
__global__ __launch_bounds__(1024) void match_scan_kernel(const int32_t* __restrict__ cnt, int n_chunks,
                                                          int32_t* __restrict__ v, int n, int n_a, int n_b,
                                                          int32_t* __restrict__ status) {
  __shared__ int wsum[16];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    int x0 = 0;
    if (i < n)
      for (int c = 0; c < n_chunks; ++c) x0 += cnt[(size_t)c * n + i];
    int x = x0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int c = carry;
    if (i < n) v[i] = c + woff + x - x0;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + woff + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int total = carry;
    v[n] = total;
    status[0] = total;
    status[1] = (total == n_a && n_a == n_b) ? 1 : 0;  // cleared by the fill pass on any off-diagonal pair
    status[2] = 0;
    status[3] = 0;
  }
}

}  // namespace mmk

using namespace mmk;

extern "C" int mmk_match_workspace_ints(int n_a, int n_b) {
  return n_a * cdiv(std::max(n_b, 1), MATCH_CHUNK) + (n_a + 1) + n_b + 1;
}

extern "C" int mmk_match_ids(const int64_t* ids_a, int n_a, const int64_t* ids_b, int n_b, int32_t* workspace,
                             int32_t* idx_a, int32_t* idx_b, int capacity, int32_t* status, void* stream) {
  MMK_REQUIRE(n_a > 0 && n_b > 0 && capacity >= 0, "empty or negative size");
  MMK_REQUIRE(ids_a && ids_b && workspace && status, "null pointer");
  MMK_REQUIRE(capacity == 0 || (idx_a && idx_b), "null index buffers");
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_MATCH, st);
  const longlong2* a = reinterpret_cast<const longlong2*>(ids_a);
  const longlong2* b = reinterpret_cast<const longlong2*>(ids_b);
  const int n_chunks = cdiv(n_b, MATCH_CHUNK);
  int32_t* cnt = workspace;                                     // [n_a * n_chunks] then row_off [n_a + 1]
  int32_t* row_off = workspace + (size_t)n_a * n_chunks;
  int32_t* cnt_b = row_off + n_a + 1;                             // [n_b]
  MMK_HIP(hipMemsetAsync(cnt_b, 0, sizeof(int32_t) * n_b, st));
  dim3 grid(cdiv(n_a, 256), n_chunks);
  hipLaunchKernelGGL((match_kernel<0>), grid, dim3(256), 0, st, a, n_a, b, n_b, n_chunks, cnt, cnt_b, nullptr, nullptr, 0, status);
  hipLaunchKernelGGL(match_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, n_chunks, row_off, n_a, n_a, n_b, status);
  hipLaunchKernelGGL((match_kernel<1>), grid, dim3(256), 0, st, a, n_a, b, n_b, n_chunks, cnt, cnt_b, idx_a, idx_b, capacity, status);
  MMK_LAUNCH_CHECK();
  return 0;
}
