// At what rate do the XCD <-> fabric links move RANDOM rows of 128 / 256 / 512 / 1024 bytes?  (round 5)
// profiles/r02_fabric_probe.txt calibrated the layer-1 aggregation (512-byte rows: 6.4-6.9 TB/s remote-only).  The two 64-float
// aggregations of the step gather 256-byte rows and move their fabric traffic at 4.25-4.74 TB/s; before building another
// kernel form for them this probe asks what the part delivers for that row size at all:
//   plain : every lane group of RB/16 lanes gathers one row per trip with a 16-byte load per lane, U trips in flight
//           (registers), T = TL + TR trips per output row (TL inside a +-64-row window = L2 hits, TR uniformly random)
//   dma   : the same trips issued as global_load ... lds (16 bytes per lane, each lane its own address, the wave's 1 KB
//           lands in lane order in a per-wave LDS ring), two output rows in flight per wave, summed from ds_read_b128
// Indices are hashed from the row id (no index traffic).  One output row of RB bytes is written per T trips.
// Build: hipcc -O3 --offload-arch=gfx950 rowsize_probe.hip -o rowsize_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
  return a;
}

__global__ void fill_kernel(float* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (float)(mix((uint32_t)i) & 0xffff) * (1.0f / 65536.0f) - 0.5f;
}

template <int RB>
__device__ __forceinline__ uint32_t pick(int r, int t, int g, int TR, int n) {
  constexpr int G = 1024 / RB;
  const uint32_t h = mix((uint32_t)r * 256u + (uint32_t)(G * t + g));
  int c;
  if (t < TR) c = (int)(h % (uint32_t)n);
  else { c = r + (int)(h % 129u) - 64; c = c < 0 ? c + n : (c >= n ? c - n : c); }
  return (uint32_t)c * (uint32_t)RB;
}

template <int RB>
__device__ __forceinline__ float4 group_sum(float4 a) {
#pragma unroll
  for (int off = RB / 16; off < 64; off <<= 1) {
    a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
  }
  return a;
}

// XCD k sweeps the k-th eighth of the output rows with all its resident waves (as csrc/spmm.hip does)
template <int RB, int U>
__global__ __launch_bounds__(256) void plain_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int TL, int TR) {
  constexpr int LPR = RB / 16;
  const int lane = threadIdx.x & 63, g = lane / LPR, li = lane % LPR;
  const int xcd = blockIdx.x & 7;
  const int waves_per_xcd = (gridDim.x / 8) * 4;
  const int wx = (blockIdx.x / 8) * 4 + (threadIdx.x >> 6);
  const int per = (n + 7) / 8;
  const int r0 = xcd * per, r1 = min(n, r0 + per);
  const char* xb = reinterpret_cast<const char*>(x);
  const int T = TL + TR;
  for (int r = r0 + wx; r < r1; r += waves_per_xcd) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = 0; t0 < T; t0 += U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = *reinterpret_cast<const float4*>(xb + (pick<RB>(r, t0 + u, g, TR, n) + (uint32_t)li * 16u));
#pragma unroll
      for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    acc = group_sum<RB>(acc);
    if (g == 0) *reinterpret_cast<float4*>(reinterpret_cast<char*>(y) + ((size_t)r * RB + li * 16u)) = acc;
  }
}

// LDS ring: T trips of one output row = T KB per wave; two output rows in flight (ring of 2 T slots).
template <int RB, int T>
__global__ __launch_bounds__(256) void dma_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int TR) {
  constexpr int LPR = RB / 16;
  extern __shared__ float4 ring[];                       // 4 waves x 2 T slots x 64 float4
  const int lane = threadIdx.x & 63, g = lane / LPR, li = lane % LPR;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7;
  const int waves_per_xcd = (gridDim.x / 8) * 4;
  const int wx = (blockIdx.x / 8) * 4 + wv;
  const int per = (n + 7) / 8;
  const int r0 = xcd * per, r1 = min(n, r0 + per);
  const char* xb = reinterpret_cast<const char*>(x);
  float4* my = ring + wv * (2 * T * 64);
  auto issue = [&](int r, int half) {
#pragma unroll
    for (int t = 0; t < T; ++t)
      __builtin_amdgcn_global_load_lds((const void*)(xb + (pick<RB>(r, t, g, TR, n) + (uint32_t)li * 16u)),
                                       (__attribute__((address_space(3))) void*)(my + (half * T + t) * 64), 16, 0, 0);
  };
  int r = r0 + wx;
  if (r >= r1) return;
  issue(r, 0);
  int half = 0;
  for (; r < r1; r += waves_per_xcd) {
    const int rn = r + waves_per_xcd;
    if (rn < r1) {
      issue(rn, half ^ 1);
      if (T == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (T == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float4 v = my[(half * T + t) * 64 + lane];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    acc = group_sum<RB>(acc);
    if (g == 0) *reinterpret_cast<float4*>(reinterpret_cast<char*>(y) + ((size_t)r * RB + li * 16u)) = acc;
    half ^= 1;
  }
}

template <typename F>
static double time_us(F launch, int reps = 20) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / reps * 1e3;
}

template <int RB>
static void run_size(float* x, float* y, int n) {
  const int mixes[][2] = {{0, 8}, {4, 4}, {5, 3}, {8, 0}};
  printf("# rows of %d bytes, table %d rows = %.1f MB, %d rows per wave-wide load\n", RB, n, (double)n * RB / 1e6, 1024 / RB);
  for (auto& m : mixes) {
    const int TL = m[0], TR = m[1];
    const double rows = (double)n * (1024 / RB) * (TL + TR), remote = (double)n * (1024 / RB) * TR;
    for (int grid : {2048, 8192}) {
      double us = time_us([&]() { hipLaunchKernelGGL((plain_kernel<RB, 4>), dim3(grid), dim3(256), 0, 0, x, y, n, TL, TR); });
      printf("plain RB=%d TL=%d TR=%d grid=%d U=4: %.1f us  gathered %.2f TB/s  remote %.2f TB/s\n", RB, TL, TR, grid, us,
             rows * RB / us / 1e6, remote * RB / us / 1e6);
      us = time_us([&]() { hipLaunchKernelGGL((plain_kernel<RB, 8>), dim3(grid), dim3(256), 0, 0, x, y, n, TL, TR); });
      printf("plain RB=%d TL=%d TR=%d grid=%d U=8: %.1f us  gathered %.2f TB/s  remote %.2f TB/s\n", RB, TL, TR, grid, us,
             rows * RB / us / 1e6, remote * RB / us / 1e6);
    }
    {
      CK(hipFuncSetAttribute((const void*)dma_kernel<RB, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16 * 1024));
      for (int grid : {512, 2048}) {
        const double us = time_us([&]() { hipLaunchKernelGGL((dma_kernel<RB, 8>), dim3(grid), dim3(256), 4 * 16 * 1024, 0, x, y, n, TR); });
        printf("dma   RB=%d TL=%d TR=%d grid=%d ring 16 KB/wave (2 blocks/CU): %.1f us  gathered %.2f TB/s  remote %.2f TB/s\n", RB, TL,
               TR, grid, us, rows * RB / us / 1e6, remote * RB / us / 1e6);
      }
    }
  }
  // 4 trips per output row: the ring of the dma form is 8 KB per wave (4 blocks of 4 waves per CU)
  for (int TR : {4, 2, 0}) {
    const int TL = 4 - TR;
    const double rows = (double)n * (1024 / RB) * 4, remote = (double)n * (1024 / RB) * TR;
    double us = time_us([&]() { hipLaunchKernelGGL((plain_kernel<RB, 4>), dim3(2048), dim3(256), 0, 0, x, y, n, TL, TR); });
    printf("plain RB=%d TL=%d TR=%d grid=2048 U=4: %.1f us  gathered %.2f TB/s  remote %.2f TB/s\n", RB, TL, TR, us,
           rows * RB / us / 1e6, remote * RB / us / 1e6);
    CK(hipFuncSetAttribute((const void*)dma_kernel<RB, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 8 * 1024));
    for (int grid : {1024, 2048}) {
      us = time_us([&]() { hipLaunchKernelGGL((dma_kernel<RB, 4>), dim3(grid), dim3(256), 4 * 8 * 1024, 0, x, y, n, TR); });
      printf("dma   RB=%d TL=%d TR=%d grid=%d ring 8 KB/wave (4-5 blocks/CU): %.1f us  gathered %.2f TB/s  remote %.2f TB/s\n", RB, TL, TR,
             grid, us, rows * RB / us / 1e6, remote * RB / us / 1e6);
    }
  }
}

int main(int argc, char** argv) {
  const int n = 235868;
  float *x, *y;
  CK(hipMalloc(&x, (size_t)n * 1024)); CK(hipMalloc(&y, (size_t)n * 1024));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, x, (size_t)n * 256);
  CK(hipDeviceSynchronize());
  const char* what = argc > 1 ? argv[1] : "all";
  const bool all = !strcmp(what, "all");
  if (all || !strcmp(what, "128")) run_size<128>(x, y, n);
  if (all || !strcmp(what, "256")) run_size<256>(x, y, n);
  if (all || !strcmp(what, "512")) run_size<512>(x, y, n);
  if (all || !strcmp(what, "1024")) run_size<1024>(x, y, n);
  return 0;
}
