// Lab: sustained rate of the matrix instructions alone (no memory traffic, operands and accumulators in registers) on all
// SIMDs, for the bf16 and the fp32 instruction, at 1 / 2 / 4 waves per SIMD and with constant vs pseudo-random operands
// (operand bit activity changes the power drawn).   hipcc -O3 --offload-arch=gfx950 mfma_rate_lab.hip -o mfma_rate_lab.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_clk[2];
template <bool BF16>
__global__ __launch_bounds__(256) void mfma_loop(float* sink, int iters, uint32_t seed) {
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  uint32_t h = (threadIdx.x + blockIdx.x * 256) * 2654435761u ^ seed;
  // four different operand pairs, one per accumulator: consecutive instructions see different operand bits
  bf16x8 a[4], b[4];
  float fa[4], fb[4];
  for (int t = 0; t < 4; ++t) {
    u32x4 pa, pb;
    for (int i = 0; i < 4; ++i) {
      h = h * 1664525u + 1013904223u;
      pa[i] = seed ? ((h & 0x7fff7fffu) | 0x3f003f00u) & 0x3fff3fffu : 0x3f803f80u;   // bf16 pairs in [0.5, 2) or 1.0
      h = h * 1664525u + 1013904223u;
      pb[i] = seed ? ((h & 0x7fff7fffu) | 0x3f003f00u) & 0x3fff3fffu : 0x3f803f80u;
    }
    a[t] = __builtin_bit_cast(bf16x8, pa); b[t] = __builtin_bit_cast(bf16x8, pb);
    fa[t] = seed ? __builtin_bit_cast(float, (pa[0] & 0x3fffffffu) | 0x3f000000u) : 1.0f;
    fb[t] = seed ? __builtin_bit_cast(float, (pb[0] & 0x3fffffffu) | 0x3f000000u) : 1.0f;
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (BF16) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[t], acc[t], 0, 0, 0);
      else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[t], acc[t], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 3 && threadIdx.x == 0) { g_clk[0] = __builtin_readcyclecounter() - c0; g_clk[1] = wall_clock64() - w0; }
}

int main() {
  float* sink;
  CK(hipMalloc(&sink, 4096 * 256 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int long_run = 0; long_run < 2; ++long_run)
  for (int random = 0; random < 2; ++random)
    for (int bf = 1; bf >= 0; --bf)
      for (int wps : {1, 2, 4}) {
        const int grid = 256 * wps, iters = (bf ? 20000 : 10000) * (long_run ? 10 : 1);          // 256 CUs x 4 SIMDs x wps waves
        auto launch = [&] {
          if (bf) hipLaunchKernelGGL(mfma_loop<true>, dim3(grid), dim3(256), 0, 0, sink, iters, random ? 12345u : 0u);
          else hipLaunchKernelGGL(mfma_loop<false>, dim3(grid), dim3(256), 0, 0, sink, iters, random ? 12345u : 0u);
        };
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 3; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = 3.0 * grid * 4.0 * iters * 4.0 * (bf ? 32768.0 : 4096.0);
        unsigned long long clk[2];
        CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk)));
        printf("%-5s %-9s operands, %d wave(s) per SIMD: %8.1f TFLOP/s  (%.0f ms of continuous matrix work, shader clock %.0f MHz)\n", bf ? "bf16" : "fp32",
               random ? "random" : "constant", wps, flop / (ms * 1e-3) / 1e12, ms, 100.0 * (double)clk[0] / (double)clk[1]);
      }
  return 0;
}
