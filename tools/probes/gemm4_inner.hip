// Probe: ceiling of a one-wave-per-SIMD inner loop with a 128 x 128 register tile per wave (256 accumulator registers),
// operands re-read from a fixed LDS image by ds_read_b128 (no global traffic, no stores): how close to the MFMA peak does
// "fragment reads in the MFMA shadow" get on gfx950?   hipcc --offload-arch=gfx950 -O3 gemm4_inner.hip -o gemm4_inner
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int VARIANT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(float* out, int nk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (256 + 256 rows) x 128 B = 128 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  for (int i = tid; i < 128 * 1024 / 4; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003c00u + (i & 7);
  __syncthreads();
  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  uint32_t offk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) offk[kk] = (uint32_t)(r * 128 + (((2 * kk + h) ^ ((r >> 1) & 7)) << 4));
  for (int kt = 0; kt < nk; ++kt) {
    const char* st = smem + (kt & 1) * 65536;
    const char* sa = st + wm * 16384;            // A rows 128 wm .. +127
    const char* sb = st + 32768 + wn * 16384;    // B rows 128 wn .. +127
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const bf16x8*>(sa + a * 4096 + offk[kk]);
#pragma unroll
      for (int b = 0; b < 4; ++b) fb[b] = *reinterpret_cast<const bf16x8*>(sb + b * 4096 + offk[kk]);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (VARIANT == 1) __builtin_amdgcn_s_barrier();   // one barrier per K step, as a ring pipeline would need
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[a][b][e];
  out[blockIdx.x * 256 + tid] = s;
}

// 8 waves (two per SIMD), 128 x 64 per wave, free running (one barrier per K step, no phases): do the two waves of a SIMD
// fill each other's fragment-read gaps?
template <int VARIANT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe8(float* out, int nk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, r = lane & 31, h = lane >> 5;
  for (int i = tid; i < 128 * 1024 / 4; i += 512) reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003c00u + (i & 7);
  __syncthreads();
  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  uint32_t offk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) offk[kk] = (uint32_t)(r * 128 + (((2 * kk + h) ^ ((r >> 1) & 7)) << 4));
  for (int kt = 0; kt < nk; ++kt) {
    const char* st = smem + (kt & 1) * 65536;
    const char* sa = st + wm * 16384;                                  // activation rows 128 wm .. +127 (4 blocks)
    const char* sb = st + 32768 + (wn >> 1) * 16384 + (wn & 1) * 8192;   // weight rows 64 wn .. +63 (2 blocks)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 fa[4], fb[2];
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const bf16x8*>(sa + a * 4096 + offk[kk]);
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[b] = *reinterpret_cast<const bf16x8*>(sb + b * 4096 + offk[kk]);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[b][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[b][a], 0, 0, 0);
    }
    if (VARIANT == 1) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[a][b][e];
  out[blockIdx.x * 512 + tid] = s;
}

// the same with NDMA LDS-DMA instructions per K step and wave (1 KiB each, re-reading a cached 64 KiB region of `src` into
// a scratch LDS area nobody reads, never waited for inside the loop): what does ISSUING the fill cost the MFMA rate?
// MODE 1: global_load_dwordx4 -> VGPR + ds_write_b128 instead of LDS-DMA (same bytes).
template <int NDMA, int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe8dma(float* out, const char* src, int nk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 128 KiB operands + 32 KiB DMA sink
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, r = lane & 31, h = lane >> 5;
  for (int i = tid; i < 128 * 1024 / 4; i += 512) reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003c00u + (i & 7);
  __syncthreads();
  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  uint32_t offk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) offk[kk] = (uint32_t)(r * 128 + (((2 * kk + h) ^ ((r >> 1) & 7)) << 4));
  const uint32_t sink = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)smem) + 128 * 1024 + wave * 4096;
  const uint32_t voff = (uint32_t)(lane * 16 + wave * 8192);
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  for (int kt = 0; kt < nk; ++kt) {
    const char* st = smem + (kt & 1) * 65536;
    const char* sa = st + wm * 16384;
    const char* sb = st + 32768 + (wn >> 1) * 16384 + (wn & 1) * 8192;
    // MODE 2 / 3: 256 KiB per workgroup (64 MiB in all: Infinity-Cache resident); MODE 4: 64 KiB per workgroup, re-read every
    // step (2 MiB per XCD: L2 resident); the per-lane offset spans 64 KiB
    const char* sbase = (MODE == 4) ? src + (size_t)blockIdx.x * 65536
                        : (MODE >= 2) ? src + (size_t)blockIdx.x * 262144 + (size_t)(kt & 3) * 65536
                                      : src + (kt & 7) * 1024;   // scalar
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 fa[4], fb[2];
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const bf16x8*>(sa + a * 4096 + offk[kk]);
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[b] = *reinterpret_cast<const bf16x8*>(sb + b * 4096 + offk[kk]);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[b][a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[b], fa[a], acc[b][a], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < NDMA / 4; ++u) {
        if (MODE == 0) {
          const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(sink + (uint32_t)((kk * (NDMA / 4) + u) & 3) * 1024u));
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(m0v));
        } else {
          u32x4 t;
          asm volatile("global_load_dwordx4 %0, %1, %2\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "v"(voff), "s"(sbase));   // worst case: waited at once
          *reinterpret_cast<u32x4*>(smem + 128 * 1024 + wave * 4096 + lane * 16 + ((kk * (NDMA / 4) + u) & 3) * 1024) = t;
        }
      }
    }
    if (MODE == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * 2 > 63 ? 63 : NDMA * 2) : "memory");   // two steps may stay in flight
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[a][b][e];
  out[blockIdx.x * 512 + tid] = s;
}

template <int NDMA, int MODE>
static void run8dma(int nk) {
  float* out;
  char* src;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipMalloc(&src, (size_t)256 * 262144 + (1 << 20));
  hipMemset(src, 0, (size_t)256 * 262144 + (1 << 20));
  auto k = probe8dma<NDMA, MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k, dim3(256), dim3(512), 160 * 1024, 0, out, src, nk);
  hipEventRecord(a);
  const int reps = 10;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(k, dim3(256), dim3(512), 160 * 1024, 0, out, src, nk);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double flops = 256.0 * 2.0 * 256 * 256 * 64 * nk * reps;
  printf("8 waves + %d %s per step and wave: %.1f us/launch, %.1f TFLOP/s\n", NDMA,
         MODE == 1 ? "global_load+ds_write (waited at once)" : MODE == 0 ? "LDS-DMA (fire and forget, one hot 8 KiB)" : MODE == 2 ? "LDS-DMA (fire and forget, 256 KiB per workgroup: Infinity-Cache-resident stream)" : MODE == 3 ? "LDS-DMA (256 KiB stream, vmcnt leaves two steps in flight)" : "LDS-DMA (fire and forget, 64 KiB per workgroup: L2-resident)",
         ms / reps * 1e3, flops / (ms * 1e-3) / 1e12);
  hipFree(out);
  hipFree(src);
}

template <int V>
static void run8(int nk) {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  auto k = probe8<V>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k, dim3(256), dim3(512), 128 * 1024, 0, out, nk);
  hipEventRecord(a);
  const int reps = 10;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(k, dim3(256), dim3(512), 128 * 1024, 0, out, nk);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double flops = 256.0 * 2.0 * 256 * 256 * 64 * nk * reps;
  printf("8 waves, variant %d nk %d: %.1f us/launch, %.1f TFLOP/s\n", V, nk, ms / reps * 1e3, flops / (ms * 1e-3) / 1e12);
  hipFree(out);
}

template <int V>
static void run(int nk) {
  float* out;
  hipMalloc(&out, 256 * 256 * sizeof(float));
  auto k = probe<V>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k, dim3(256), dim3(256), 128 * 1024, 0, out, nk);
  hipEventRecord(a);
  const int reps = 10;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(k, dim3(256), dim3(256), 128 * 1024, 0, out, nk);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double flops = 256.0 * 2.0 * 256 * 256 * 64 * nk * reps;
  printf("variant %d nk %d: %.1f us/launch, %.1f TFLOP/s\n", V, nk, ms / reps * 1e3, flops / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  run<0>(4096);
  run<1>(4096);
  run8<0>(4096);
  run8<1>(4096);
  run8dma<4, 0>(2048);
  run8dma<8, 0>(2048);
  run8dma<16, 0>(2048);
  run8dma<8, 1>(2048);
  run8dma<8, 2>(2048);
  run8dma<8, 3>(2048);
  run8dma<8, 4>(2048);
  run8dma<16, 4>(2048);
  return 0;
}
