// Upper bound for the SpMM's gather path: rows of 512 B (d = 128 fp32) gathered with indices that cost nothing
// (hashed from the row id, inside a window of +-W rows), DEG neighbours per output row, two neighbours per
// wave-wide 16-byte load (half-wave per row), U loads in flight.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
  return a;
}

template <int U>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ x, float* __restrict__ y, int n,
                                                     int deg, int window, int rows_per_wave) {
  extern __shared__ float occupancy_ballast[];     // dynamic LDS only limits the resident blocks per CU
  if (rows_per_wave < 0) occupancy_ballast[threadIdx.x] = 0.f;
  const int lane = threadIdx.x & 63, g = lane >> 5, li = lane & 31;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7;
  // XCD-contiguous row ranges like the real kernel
  const int waves_per_xcd = (gridDim.x / 8) * 4;
  const int wx = (blockIdx.x / 8) * 4 + (threadIdx.x >> 6);
  const int per = (n + 7) / 8;
  const int r0 = xcd * per, r1 = min(n, r0 + per);
  const char* xb = reinterpret_cast<const char*>(x);
  for (int r = r0 + wx; r < r1; r += waves_per_xcd) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = 0; t0 < deg; t0 += 2 * U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t h = mix((uint32_t)r * 64u + (uint32_t)(t0 + 2 * u + g));
        int c = r + (int)(h % (uint32_t)(2 * window + 1)) - window;
        c = c < 0 ? c + n : (c >= n ? c - n : c);
        v[u] = *reinterpret_cast<const float4*>(xb + ((uint32_t)c * 512u + (uint32_t)li * 16u));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32);
    acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
    if (g == 0) *reinterpret_cast<float4*>(reinterpret_cast<char*>(y) + ((size_t)r * 512u + li * 16u)) = acc;
  }
}

int main(int argc, char** argv) {
  const int n = 235868, d = 128;
  float *x, *y;
  hipMalloc(&x, (size_t)n * d * 4); hipMalloc(&y, (size_t)n * d * 4);
  hipMemset(x, 0, (size_t)n * d * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  if (argc > 1) {
    // occupancy study: how the gather rate holds up with few resident waves per CU and deep unrolling
    // (what a kernel that also keeps a 64 KB weight image in LDS could afford)
    const int deg = 16;
    for (int w : {64, 117000}) for (int lds_kb : {0, 40, 64}) for (int U : {4, 8}) {
      const int grid = 2048;
      const size_t lds = (size_t)lds_kb * 1024;
      auto launch = [&]() {
        if (U == 4) hipLaunchKernelGGL(gather_kernel<4>, dim3(grid), dim3(256), lds, 0, x, y, n, deg, w, 0);
        else hipLaunchKernelGGL(gather_kernel<8>, dim3(grid), dim3(256), lds, 0, x, y, n, deg, w, 0);
      };
      for (int i = 0; i < 3; ++i) launch();
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms / 20 * 1e3, gathered = (double)n * deg * 512;
      const int blocks_per_cu = lds_kb ? (160 / lds_kb < 8 ? 160 / lds_kb : 8) : 8;
      printf("deg=%d window=%d lds=%dKB (%d waves/CU) U=%d: %.1f us  gather %.1f TB/s\n", deg, w, lds_kb,
             blocks_per_cu * 4, U, us, gathered / us / 1e6);
    }
    return 0;
  }
  const int degs[] = {8, 16};
  const int windows[] = {64, 4096, 117000};
  const int grids[] = {2048, 8192, 16384};
  for (int deg : degs) for (int w : windows) for (int grid : grids) for (int U : {4, 8}) {
    if (U * 2 > deg) continue;
    auto launch = [&]() {
      if (U == 4) hipLaunchKernelGGL(gather_kernel<4>, dim3(grid), dim3(256), 0, 0, x, y, n, deg, w, 0);
      else hipLaunchKernelGGL(gather_kernel<8>, dim3(grid), dim3(256), 0, 0, x, y, n, deg, w, 0);
    };
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms / 20 * 1e3, gathered = (double)n * deg * 512;
    printf("deg=%d window=%d grid=%d U=%d: %.1f us  gather %.1f TB/s  (%.1f ps per gathered row)\n", deg, w, grid, U, us,
           gathered / us / 1e6, us * 1e6 / ((double)n * deg));
  }
  return 0;
}
