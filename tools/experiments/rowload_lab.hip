// Lab: what does the SHAPE of a wave's 16-byte loads cost when a [M, K] fp32 matrix is streamed once, 32 rows per wave
// at a time (the row GEMMs' operand stream)?  Same bytes, same number of load instructions, three lane -> address maps:
//   A  row per lane, 32 rows x 2 pieces of 16 bytes per instruction (rows_gemm.hip / gemm_ktile.hip today: what the
//      32x32 matrix instructions' operand layout asks for)
//   B  16 rows x 64 contiguous bytes per instruction (what the 16x16 instructions' layout would ask for)
//   C  2 rows x 512 contiguous bytes per instruction (fully coalesced: needs a transpose through LDS afterwards)
// hipcc -O3 --offload-arch=gfx950 rowload_lab.hip -o rowload_lab.bin ; ./rowload_lab.bin M K
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int PAT>
__global__ __launch_bounds__(512, 2) void stream_kernel(const float* __restrict__ x, int m, int k, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 512 + threadIdx.x) >> 6, n_waves = gridDim.x * 8;
  const int n_tiles = (m + 31) / 32, f4_per_row = k / 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int tile = wave; tile < n_tiles; tile += n_waves) {
    // a 128-byte column block (8 float4) of the tile's 32 rows = 4 load instructions in every pattern
    for (int cb = 0; cb < f4_per_row / 8; ++cb) {
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int row, c4;
        if (PAT == 0) { row = lane & 31; c4 = (lane >> 5) * 4 + i; }                       // A
        else if (PAT == 1) { row = (lane & 15) + 16 * (i >> 1); c4 = (lane >> 4) + 4 * (i & 1); }   // B
        else { row = 8 * i + (lane >> 3); c4 = lane & 7; }                                   // C (8 rows x 128 B per instruction here)
        row = min(tile * 32 + row, m - 1);
        v[i] = reinterpret_cast<const float4*>(x + (int64_t)row * k)[cb * 8 + c4];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
  }
  sink[blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 19793, k = argc > 2 ? atoi(argv[2]) : 8736, reps = 10;
  float *x, *sink;
  CK(hipMalloc(&x, (size_t)m * k * 4));
  CK(hipMalloc(&sink, 512 * 512 * 4));
  CK(hipMemset(x, 0, (size_t)m * k * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto kern, int grid) {
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, x, m, k, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, x, m, k, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps;
    printf("%-56s grid %4d  %8.1f us  %6.0f GB/s\n", name, grid, us, (double)m * k * 4 / us / 1e3);
  };
  printf("M=%d K=%d (%.0f MB)\n", m, k, (double)m * k * 4 / 1e6);
  for (int grid : {256, 512}) {
    run("A: 32 rows x 2 x 16 B per instruction (today)", stream_kernel<0>, grid);
    run("B: 16 rows x 64 B per instruction", stream_kernel<1>, grid);
    run("C: 8 rows x 128 B per instruction", stream_kernel<2>, grid);
  }
  return 0;
}
