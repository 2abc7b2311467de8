// Device-side example-id matching for gfx950.
//
// Replaces find_matching_indices (mmlearn/datasets/core/example.py:160-166): the reference
// broadcasts [N,1,2]==[1,M,2] into an N*M*2 bool tensor, reduces, and calls torch.where
// (nonzero + host sync).  Here b is taken in chunks of 1024 ids; a workgroup holds a chunk in LDS together with an
// open-addressing hash table over it (4096 slots), and every row of a probes the table: O(N M / 1024) probes instead of
// N M compares.  A chunk that holds the same id twice cannot be served by one probe; its workgroup then compares against
// the whole chunk (32-bit hashes first, eight candidates per two broadcast ds_read_b128, exact 2 x int64 compare on a hash
// hit).  Output order is row-major (i ascending, then j ascending), exactly torch.where's, and independent of scheduling
// (integer atomics only feed counters / flags / the hash table, whose layout does not affect the result).
//
//   small problems (n_a, n_b <= 2048): ONE launch of one 1024-thread workgroup: count sweep -> block scan -> fill
//   otherwise: count pass over a (256 rows) x (1024-id chunk) grid -> single-block scan of the row totals -> fill pass
#include <algorithm>

#include "common.h"

namespace mmk {

constexpr int MATCH_CHUNK = 1024;   // ids of b per LDS chunk (16 KiB of ids + 4 KiB of hashes)
constexpr int MATCH_SMALL = 2048;   // single-workgroup path up to this many rows on either side

__device__ __forceinline__ uint32_t id_hash(const longlong2 v) {
  const uint64_t x = (uint64_t)v.x, y = (uint64_t)v.y;
  uint32_t h = (uint32_t)y ^ ((uint32_t)(y >> 32) * 0x85EBCA6Bu) ^ ((uint32_t)x * 0x9E3779B1u) ^ ((uint32_t)(x >> 32) * 0xC2B2AE35u);
  // avalanche: example indices are consecutive integers, and consecutive table slots would turn linear probing into a
  // walk over the whole run (measured: 160 us instead of 6 for 8192 x 8192 ids)
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}

// cooperative load of b[j0 .. j0 + len) and its hashes; hash slots up to the next multiple of 8 are zeroed (a spurious
// hash hit there is rejected by the bounds check of the exact compare)
__device__ __forceinline__ void load_chunk(const longlong2* __restrict__ b, int j0, int len, longlong2* sb, uint32_t* sh, int tid,
                                           int nthreads) {
  for (int t = tid; t < ((len + 7) & ~7); t += nthreads) {
    if (t < len) {
      const longlong2 v = b[j0 + t];
      sb[t] = v;
      sh[t] = id_hash(v);
    } else {
      sh[t] = 0u;
    }
  }
}

// Open-addressing table over the chunk in LDS: table[slot] = chunk-local index or -1.  Returns through *dup whether the
// chunk holds an id twice (the table is then incomplete and must not be probed).  Ends with a barrier.
constexpr int MATCH_SLOTS = 4 * MATCH_CHUNK;   // load factor <= 0.25: short probe chains
__device__ __forceinline__ void build_table(const longlong2* sb, const uint32_t* sh, int len, int* table, int* dup, int tid, int nthreads) {
  for (int t = tid; t < MATCH_SLOTS; t += nthreads) table[t] = -1;
  if (tid == 0) *dup = 0;
  __syncthreads();
  for (int t = tid; t < len; t += nthreads) {
    const longlong2 v = sb[t];
    int slot = (int)(sh[t] & (MATCH_SLOTS - 1));
    for (int probe = 0; probe < MATCH_SLOTS; ++probe) {   // <= len <= MATCH_CHUNK entries: a free slot always exists
      const int prev = atomicCAS(&table[slot], -1, t);
      if (prev == -1) break;
      const longlong2 o = sb[prev];
      if (o.x == v.x && o.y == v.y) {
        *dup = 1;
        break;
      }
      slot = (slot + 1) & (MATCH_SLOTS - 1);
    }
  }
  __syncthreads();
}
// chunk-local index of the (only) entry equal to `mine`, or -1
__device__ __forceinline__ int probe_table(const longlong2 mine, uint32_t mh, const longlong2* sb, const int* table) {
  int slot = (int)(mh & (MATCH_SLOTS - 1));
  for (int probe = 0; probe < MATCH_SLOTS; ++probe) {
    const int t = table[slot];
    if (t < 0) return -1;
    const longlong2 o = sb[t];
    if (o.x == mine.x && o.y == mine.y) return t;
    slot = (slot + 1) & (MATCH_SLOTS - 1);
  }
  return -1;
}

// visits every j (ascending, chunk-local index t) with sb[t] == mine
template <typename F>
__device__ __forceinline__ void for_each_match(const longlong2 mine, uint32_t mh, const longlong2* sb, const uint32_t* sh, int len,
                                               F&& on_match) {
  for (int t0 = 0; t0 < len; t0 += 8) {
    const uint4 h0 = *reinterpret_cast<const uint4*>(sh + t0), h1 = *reinterpret_cast<const uint4*>(sh + t0 + 4);
    const bool any = (h0.x == mh) | (h0.y == mh) | (h0.z == mh) | (h0.w == mh) | (h1.x == mh) | (h1.y == mh) | (h1.z == mh) | (h1.w == mh);
    if (any) {
      for (int e = 0; e < 8; ++e) {
        const int t = t0 + e;
        if (t < len) {
          const longlong2 o = sb[t];
          if (o.x == mine.x && o.y == mine.y) on_match(t);
        }
      }
    }
  }
}

// the matches of `mine` in the chunk, ascending: one probe, or the full compare when the chunk repeats an id
template <typename F>
__device__ __forceinline__ void visit_matches(const longlong2 mine, uint32_t mh, const longlong2* sb, const uint32_t* sh, const int* table,
                                              bool dup, int len, F&& on_match) {
  if (!dup) {
    const int t = probe_table(mine, mh, sb, table);
    if (t >= 0) on_match(t);
  } else {
    for_each_match(mine, mh, sb, sh, len, on_match);
  }
}

// ------------------------------------------------------------------ small problems: one workgroup, one launch
__global__ __launch_bounds__(1024) void match_small_kernel(const longlong2* __restrict__ a, int n_a, const longlong2* __restrict__ b,
                                                           int n_b, int32_t* __restrict__ idx_a, int32_t* __restrict__ idx_b,
                                                           int capacity, int32_t* __restrict__ status) {
  __shared__ longlong2 sb[MATCH_CHUNK];
  __shared__ __attribute__((aligned(16))) uint32_t sh[MATCH_CHUNK];
  __shared__ int cnt_b[MATCH_SMALL];
  __shared__ int table[MATCH_SLOTS];
  __shared__ int dup;
  __shared__ int wsum[16];
  __shared__ int flags[4];   // 0: some row has several matches, 1: some pair is off the diagonal, 2: some column has several
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int t = tid; t < n_b; t += 1024) cnt_b[t] = 0;
  if (tid < 4) flags[tid] = 0;
  constexpr int SLOTS = MATCH_SMALL / 1024;
  longlong2 mine[SLOTS];
  uint32_t mh[SLOTS];
  int cnt[SLOTS], first[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int i = s * 1024 + tid;
    mine[s] = i < n_a ? a[i] : make_longlong2(0, 0);
    mh[s] = id_hash(mine[s]);
    cnt[s] = 0;
    first[s] = -1;
  }
  const int n_chunks = (n_b + MATCH_CHUNK - 1) / MATCH_CHUNK;
  for (int c = 0; c < n_chunks; ++c) {
    const int j0 = c * MATCH_CHUNK, len = min(MATCH_CHUNK, n_b - j0);
    __syncthreads();
    load_chunk(b, j0, len, sb, sh, tid, 1024);
    __syncthreads();
    build_table(sb, sh, len, table, &dup, tid, 1024);
    const bool chunk_dup = dup != 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      if (s * 1024 + tid < n_a) {
        visit_matches(mine[s], mh[s], sb, sh, table, chunk_dup, len, [&](int t) {
          if (cnt[s] == 0) first[s] = j0 + t;
          ++cnt[s];
          atomicAdd(&cnt_b[j0 + t], 1);
        });
      }
    }
  }
  // exclusive scan of the row totals in row order (slot 0 = rows 0..1023, slot 1 = rows 1024..2047)
  int pos[SLOTS];
  int carry = 0;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    int x = cnt[s];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
      if (w < wave) woff += wsum[w];
      tot += wsum[w];
    }
    pos[s] = carry + woff + x - cnt[s];
    carry += tot;
  }
  const int total = carry;
  bool multi = false, off_diag = false;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int i = s * 1024 + tid;
    if (cnt[s] == 1) {
      if (pos[s] < capacity) {
        idx_a[pos[s]] = i;
        idx_b[pos[s]] = first[s];
      }
      if (pos[s] != i || first[s] != i) off_diag = true;
    } else if (cnt[s] > 1) {
      multi = true;
      off_diag = true;
    }
  }
  if (multi) flags[0] = 1;
  if (off_diag) flags[1] = 1;
  __syncthreads();   // cnt_b complete (LDS atomics of the sweep), flags[0] visible
  for (int t = tid; t < n_b; t += 1024)
    if (cnt_b[t] > 1) flags[2] = 1;
  if (flags[0]) {
    // rows in several pairs: second sweep in chunk order, only those rows write (uniform branch: flags[0] is block-wide)
    for (int c = 0; c < n_chunks; ++c) {
      const int j0 = c * MATCH_CHUNK, len = min(MATCH_CHUNK, n_b - j0);
      __syncthreads();
      load_chunk(b, j0, len, sb, sh, tid, 1024);
      __syncthreads();
      build_table(sb, sh, len, table, &dup, tid, 1024);
      const bool chunk_dup = dup != 0;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        if (cnt[s] > 1) {
          visit_matches(mine[s], mh[s], sb, sh, table, chunk_dup, len, [&](int t) {
            if (pos[s] < capacity) {
              idx_a[pos[s]] = s * 1024 + tid;
              idx_b[pos[s]] = j0 + t;
            }
            ++pos[s];
          });
        }
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    status[0] = total;
    status[1] = (total == n_a && n_a == n_b && !flags[1]) ? 1 : 0;
    status[2] = flags[0];
    status[3] = flags[2];
  }
}

// ------------------------------------------------------------------ large problems: count -> scan -> fill
// MODE 0: cnt[chunk*n_a + i] = #{j in chunk : a[i] == b[j]}, cnt_b[j] += 1 per match.
// MODE 1: write the pairs of (i, chunk) behind row_off[i] + the row's earlier chunks; maintain the status flags.
template <int MODE>
__global__ __launch_bounds__(256) void match_kernel(const longlong2* __restrict__ a, int n_a, const longlong2* __restrict__ b,
                                                    int n_b, int n_chunks, int32_t* __restrict__ cnt, int32_t* __restrict__ row_tot,
                                                    int32_t* __restrict__ cnt_b, int32_t* __restrict__ idx_a,
                                                    int32_t* __restrict__ idx_b, int capacity, int32_t* __restrict__ status) {
  __shared__ longlong2 sb[MATCH_CHUNK];
  __shared__ __attribute__((aligned(16))) uint32_t sh[MATCH_CHUNK];
  __shared__ int table[MATCH_SLOTS];
  __shared__ int dup;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int chunk = blockIdx.y;
  const int j0 = chunk * MATCH_CHUNK;
  const int len = min(MATCH_CHUNK, n_b - j0);
  load_chunk(b, j0, len, sb, sh, threadIdx.x, 256);
  __syncthreads();
  build_table(sb, sh, len, table, &dup, threadIdx.x, 256);
  const bool chunk_dup = dup != 0;
  if (i >= n_a) return;
  const longlong2 mine = a[i];
  const uint32_t mh = id_hash(mine);
  if (MODE == 0) {
    int c = 0;
    visit_matches(mine, mh, sb, sh, table, chunk_dup, len, [&](int t) {
      ++c;
      atomicAdd(&cnt_b[j0 + t], 1);   // rare: feeds the "column repeats" flag
    });
    cnt[(size_t)chunk * n_a + i] = c;  // [chunk][row]: coalesced for the fill pass
    if (c) atomicAdd(&row_tot[i], c);  // integer: order-independent
  } else {
    if (cnt[(size_t)chunk * n_a + i] == 0) {
      if (chunk == 0) {
        const int32_t* row_off = cnt + (size_t)n_a * n_chunks;
        if (row_off[i + 1] - row_off[i] > 1) status[2] = 1;  // row i is in several pairs
      }
      return;
    }
    const int32_t* row_off = cnt + (size_t)n_a * n_chunks;
    int pos = row_off[i];
    if (row_off[i + 1] - pos > 1) status[2] = 1;
    for (int c = 0; c < chunk; ++c) pos += cnt[(size_t)c * n_a + i];
    bool off_diag = false, col_rep = false;
    visit_matches(mine, mh, sb, sh, table, chunk_dup, len, [&](int t) {   // ascending j: row-major order
      const int j = j0 + t;
      if (pos < capacity) {
        idx_a[pos] = i;
        idx_b[pos] = j;
      }
      if (pos != i || j != i) off_diag = true;
      if (cnt_b[j] > 1) col_rep = true;
      ++pos;
    });
    if (off_diag) status[1] = 0;
    if (col_rep) status[3] = 1;
  }
}

// row_off[0..n] = exclusive scan of the per-row totals; status = {total, identity candidate, 0, 0}.  One workgroup;
// thread t owns the `per` (multiple of 4) contiguous rows from t * per: 16-byte loads, one block scan, 16-byte-free stores.
__global__ __launch_bounds__(1024) void match_scan_kernel(const int32_t* __restrict__ row_tot, int per, int32_t* __restrict__ v, int n,
                                                          int n_a, int n_b, int32_t* __restrict__ status) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int4* mine = reinterpret_cast<const int4*>(row_tot + (size_t)tid * per);   // row_tot holds 1024 * per entries (zero padded)
  int tot = 0;
  for (int q = 0; q < per / 4; ++q) {
    const int4 t4 = mine[q];
    tot += t4.x + t4.y + t4.z + t4.w;
  }
  int x = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  int woff = 0, total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) woff += wsum[w];
    total += wsum[w];
  }
  int run = woff + x - tot;
  for (int q = 0; q < per / 4; ++q) {
    const int4 t4 = mine[q];
    const int i = tid * per + 4 * q;
    if (i < n) v[i] = run;
    run += t4.x;
    if (i + 1 < n) v[i + 1] = run;
    run += t4.y;
    if (i + 2 < n) v[i + 2] = run;
    run += t4.z;
    if (i + 3 < n) v[i + 3] = run;
    run += t4.w;
  }
  if (tid == 0) {
    v[n] = total;
    status[0] = total;
    status[1] = (total == n_a && n_a == n_b) ? 1 : 0;  // cleared by the fill pass on any off-diagonal pair
    status[2] = 0;
    status[3] = 0;
  }
}

}  // namespace mmk

using namespace mmk;

static int scan_rows_per_thread(int n_a) { return round_up(cdiv(n_a, 1024), 4); }

extern "C" int mmk_match_workspace_ints(int n_a, int n_b) {
  // cnt [n_a * n_chunks] | row_off [n_a + 1] | (pad to 4) | row_tot [1024 * per] | cnt_b [n_b]
  return n_a * cdiv(std::max(n_b, 1), MATCH_CHUNK) + (n_a + 1) + 3 + 1024 * scan_rows_per_thread(n_a) + n_b + 1;
}

extern "C" int mmk_match_ids(const int64_t* ids_a, int n_a, const int64_t* ids_b, int n_b, int32_t* workspace,
                             int32_t* idx_a, int32_t* idx_b, int capacity, int32_t* status, void* stream) {
  MMK_REQUIRE(n_a > 0 && n_b > 0 && capacity >= 0, "empty or negative size");
  MMK_REQUIRE(ids_a && ids_b && workspace && status, "null pointer");
  MMK_REQUIRE(capacity == 0 || (idx_a && idx_b), "null index buffers");
  MMK_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "workspace must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  ProfScope ps(MMK_K_MATCH, st);
  const longlong2* a = reinterpret_cast<const longlong2*>(ids_a);
  const longlong2* b = reinterpret_cast<const longlong2*>(ids_b);
  if (n_a <= MATCH_SMALL && n_b <= MATCH_SMALL) {
    hipLaunchKernelGGL(match_small_kernel, dim3(1), dim3(1024), 0, st, a, n_a, b, n_b, idx_a, idx_b, capacity, status);
    MMK_LAUNCH_CHECK();
    return 0;
  }
  const int n_chunks = cdiv(n_b, MATCH_CHUNK);
  const int per = scan_rows_per_thread(n_a);
  int32_t* cnt = workspace;                                     // [n_a * n_chunks] then row_off [n_a + 1]
  int32_t* row_off = workspace + (size_t)n_a * n_chunks;
  int32_t* row_tot = workspace + round_up((int)((size_t)n_a * n_chunks + n_a + 1), 4);   // [1024 * per], 16-byte aligned
  int32_t* cnt_b = row_tot + 1024 * per;                        // [n_b]
  MMK_HIP(hipMemsetAsync(row_tot, 0, sizeof(int32_t) * ((size_t)1024 * per + n_b), st));
  dim3 grid(cdiv(n_a, 256), n_chunks);
  hipLaunchKernelGGL((match_kernel<0>), grid, dim3(256), 0, st, a, n_a, b, n_b, n_chunks, cnt, row_tot, cnt_b, nullptr, nullptr, 0, status);
  hipLaunchKernelGGL(match_scan_kernel, dim3(1), dim3(1024), 0, st, row_tot, per, row_off, n_a, n_a, n_b, status);
  hipLaunchKernelGGL((match_kernel<1>), grid, dim3(256), 0, st, a, n_a, b, n_b, n_chunks, cnt, row_tot, cnt_b, idx_a, idx_b, capacity, status);
  MMK_LAUNCH_CHECK();
  return 0;
}
