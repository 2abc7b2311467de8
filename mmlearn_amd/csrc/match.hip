// Device-side example-id matching for gfx950.
//
// Replaces find_matching_indices (mmlearn/datasets/core/example.py:160-166): the reference
// broadcasts [N,1,2]==[1,M,2] into an N*M*2 bool tensor, reduces, and calls torch.where
// (nonzero + host sync).  Here: count pass -> single-block exclusive scan -> fill pass, the
// (dataset_index, example_index) rows of b streamed through LDS and broadcast to all lanes.
// Output order is row-major (i ascending, then j ascending), exactly torch.where's.
#include "common.h"

namespace mmk {

constexpr int MATCH_CHUNK = 1024;  // id pairs of b per LDS refill (16 KiB)

// MODE 0: counts[i] = #{j : a[i] == b[j]}.   MODE 1: write pairs at offs[i]...
template <int MODE>
__global__ __launch_bounds__(256) void match_kernel(const longlong2* __restrict__ a, int n_a, const longlong2* __restrict__ b,
                                                    int n_b, int32_t* __restrict__ counts, const int32_t* __restrict__ offs,
                                                    int32_t* __restrict__ idx_a, int32_t* __restrict__ idx_b, int capacity,
                                                    int32_t* __restrict__ status) {
  __shared__ longlong2 sb[MATCH_CHUNK];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < n_a;
  longlong2 mine = make_longlong2(0, 0);
  if (live) mine = a[i];
  int cnt = 0;
  int pos = (MODE == 1 && live) ? offs[i] : 0;
  bool not_identity = false;
  for (int j0 = 0; j0 < n_b; j0 += MATCH_CHUNK) {
    const int len = min(MATCH_CHUNK, n_b - j0);
    __syncthreads();
    for (int t = threadIdx.x; t < len; t += 256) sb[t] = b[j0 + t];
    __syncthreads();
    if (live) {
      for (int t = 0; t < len; ++t) {
        const longlong2 o = sb[t];
        if (o.x == mine.x && o.y == mine.y) {
          if (MODE == 0) {
            ++cnt;
          } else {
            if (pos < capacity) {
              idx_a[pos] = i;
              idx_b[pos] = j0 + t;
            }
            if (pos != i || j0 + t != i) not_identity = true;
            ++pos;
          }
        }
      }
    }
  }
  if (MODE == 0) {
    if (live) counts[i] = cnt;
  } else {
    if (not_identity) status[1] = 0;
  }
}

// exclusive scan of counts[0..n) in place, counts[n] = total; flags any count > 1 into *dup
__global__ __launch_bounds__(1024) void scan_kernel(int32_t* __restrict__ counts, int n, int32_t* dup) {
  __shared__ int wsum[16];
  __shared__ int carry;
  __shared__ int any_dup;
  if (threadIdx.x == 0) {
    carry = 0;
    any_dup = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < n ? counts[i] : 0;
    if (v > 1) any_dup = 1;
    int x = v;
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
    if (i < n) counts[i] = c + woff + x - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + woff + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counts[n] = carry;
    *dup = any_dup;
  }
}

__global__ void match_status_kernel(const int32_t* offs_a, int n_a, int n_b, int32_t* status) {
  const int total = offs_a[n_a];
  status[0] = total;
  status[1] = (total == n_a && n_a == n_b) ? 1 : 0;  // cleared by the fill pass on any off-diagonal pair
}

}  // namespace mmk

using namespace mmk;

extern "C" int mmk_match_ids(const int64_t* ids_a, int n_a, const int64_t* ids_b, int n_b, int32_t* row_count,
                             int32_t* idx_a, int32_t* idx_b, int capacity, int32_t* status, void* stream) {
  MMK_REQUIRE(n_a >= 0 && n_b >= 0 && capacity >= 0, "negative size");
  MMK_REQUIRE(row_count && status, "null workspace");
  MMK_REQUIRE(capacity == 0 || (idx_a && idx_b), "null index buffers");
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_MATCH, st);
  const longlong2* a = reinterpret_cast<const longlong2*>(ids_a);
  const longlong2* b = reinterpret_cast<const longlong2*>(ids_b);
  int32_t* cnt_a = row_count;            // [n_a + 1]
  int32_t* cnt_b = row_count + n_a + 1;  // [n_b + 1]
  if (n_a > 0) {
    hipLaunchKernelGGL((match_kernel<0>), dim3(cdiv(n_a, 256)), dim3(256), 0, st, a, n_a, b, n_b, cnt_a, nullptr, nullptr,
                       nullptr, 0, status);
    MMK_LAUNCH_CHECK();
  }
  if (n_b > 0) {  // column multiplicities (only for the "idx_b repeats" flag)
    hipLaunchKernelGGL((match_kernel<0>), dim3(cdiv(n_b, 256)), dim3(256), 0, st, b, n_b, a, n_a, cnt_b, nullptr, nullptr,
                       nullptr, 0, status);
    MMK_LAUNCH_CHECK();
  }
  // a row of a that matches k > 1 rows of b repeats in idx_a; likewise for b
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, cnt_a, n_a, status + 2);
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, cnt_b, n_b, status + 3);
  hipLaunchKernelGGL(match_status_kernel, dim3(1), dim3(1), 0, st, cnt_a, n_a, n_b, status);
  MMK_LAUNCH_CHECK();
  if (n_a > 0 && n_b > 0) {
    hipLaunchKernelGGL((match_kernel<1>), dim3(cdiv(n_a, 256)), dim3(256), 0, st, a, n_a, b, n_b, nullptr, cnt_a, idx_a, idx_b,
                       capacity, status);
    MMK_LAUNCH_CHECK();
  }
  return 0;
}
