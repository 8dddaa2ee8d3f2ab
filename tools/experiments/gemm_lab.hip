// Stand-alone lab for the MFMA row kernels at the bench shapes (N = 235,868 rows, S1 = 178,921 Del rows, widths
// 128 / 64).  Includes the PRODUCTION source (csrc/rows_gemm.hip) so that variants are compared against exactly
// what the library runs.  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -I../../gnndelete_amd/csrc
#include "../../gnndelete_amd/csrc/rows_gemm.hip"

#include <algorithm>
#include <vector>

namespace gd { char* error_buffer() { static thread_local char b[256]; return b; } }
extern "C" int gd_adam_at_f32(float*, const float*, float*, float*, const int32_t*, int64_t, double, double, double, double, void*) { return 0; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <typename F> static double time_us(F launch, int reps = 20) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / reps * 1e3;
}

__global__ void fill_kernel(float* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t a = (uint32_t)i * 2654435761u + seed; a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15;
    p[i] = (float)(a & 0xffff) * (1.0f / 65536.0f) - 0.5f;
  }
}
static float* dev_rand(size_t n, uint32_t seed) {
  float* p; CK(hipMalloc(&p, n * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, p, n, seed);
  return p;
}

#include "gemm_lab_variants.h"

int main(int argc, char** argv) {
  const int n = 235868, s1 = 178921;
  std::vector<int32_t> idx;
  { uint32_t s = 7; for (int i = 0; i < n && (int)idx.size() < s1; ++i) { s = s * 1664525u + 1013904223u; if ((s >> 8) % 1000 < 760 || n - i <= s1 - (int)idx.size()) idx.push_back(i); } }
  int32_t* d_idx; CK(hipMalloc(&d_idx, idx.size() * 4)); CK(hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
  float *x = dev_rand((size_t)n * 128, 1), *y = dev_rand((size_t)n * 128, 2), *z = dev_rand((size_t)n * 128, 3),
        *tm = dev_rand((size_t)n * 128, 4), *dh = dev_rand((size_t)n * 128, 5), *w = dev_rand(128 * 128, 6);
  uint32_t* bits; CK(hipMalloc(&bits, (size_t)n * 4 * 4)); CK(hipMemset(bits, 0xff, (size_t)n * 16));
  CK(hipDeviceSynchronize());
  auto report = [&](const char* name, double us, double flops, double bytes) {
    printf("%-78s %7.1f us  %6.1f TF (%.2f of 157.3)  %6.0f GB/s\n", name, us, flops / us / 1e6, flops / us / 1e6 / 157.3, bytes / us / 1e3);
  };
  // ---- forward kernels
  report("stage 1  x W1^T            dense 235,868 x 128 -> 128", time_us([&] { gd_rows_gemm_f32(x, 128, nullptr, n, w, 128, 128, 1, nullptr, 0, y, 128, nullptr, nullptr); }),
         2.0 * n * 128 * 128, 8.0 * n * 128);
  report("stage 3  Del-1 (+ signs)   gathered 178,921 x 128 -> 128", time_us([&] { gd_rows_gemm_signs_f32(x, 128, d_idx, s1, w, 128, 128, 0, nullptr, 0, y, 128, nullptr, bits, nullptr); }),
         2.0 * s1 * 128 * 128, 8.0 * s1 * 128);
  {
    uint8_t* sel; CK(hipMalloc(&sel, n)); CK(hipMemset(sel, 1, n));
    report("stage 5  relu(z1|pre1) W2^T dense select 235,868 x 128 -> 64", time_us([&] { gd_rows_gemm_select_f32(x, z, sel, 128, nullptr, n, w, 128, 64, 1, nullptr, 1, y, 64, nullptr); }),
           2.0 * n * 128 * 64, 4.0 * n * (128 + 64));
  }
  report("stage 10 dh = (dt2 W2) gated gathered 178,921 x 64 -> 128", time_us([&] { gd_rows_gemm_gated_f32(x, 64, d_idx, s1, w, 64, 128, 0, bits, y, 128, nullptr); }),
         2.0 * s1 * 128 * 64, 4.0 * s1 * (128 + 64));
  // ---- weight gradient
  const int64_t ws_n = gd_rows_gemm_wgrad_workspace(s1, 128, 128);
  float *ws = dev_rand(ws_n, 8), *dw = dev_rand(128 * 128, 9);
  report("stage 4  wgrad plain       a[S1]^T g[S1] (2 operands)", time_us([&] { gd_rows_gemm_wgrad_f32(x, 128, d_idx, z, 128, d_idx, nullptr, nullptr, s1, 128, 128, dw, 0, ws, nullptr); }),
         2.0 * s1 * 128 * 128, 8.0 * s1 * 128);
  report("stage 4  wgrad + g_add     (3 operands)", time_us([&] { gd_rows_gemm_wgrad_f32(x, 128, d_idx, z, 128, d_idx, nullptr, dh, s1, 128, 128, dw, 0, ws, nullptr); }),
         2.0 * s1 * 128 * 128, 12.0 * s1 * 128);
  {
    std::vector<int32_t> slot(s1); for (int i = 0; i < s1; ++i) slot[i] = i;
    std::vector<float> ones(s1, 1.0f);
    int32_t* d_slot; float *coef, *cnt, *lp;
    CK(hipMalloc(&d_slot, s1 * 4)); CK(hipMemcpy(d_slot, slot.data(), s1 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&coef, s1 * 4)); CK(hipMemcpy(coef, ones.data(), s1 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&cnt, s1 * 4)); CK(hipMemcpy(cnt, ones.data(), s1 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&lp, 2 * 4096 * 4));
    report("stage 4  wgrad + loss + g_add (4 operands, what the step runs)", time_us([&] {
             gd_rows_gemm_wgrad_loss_f32(x, 128, d_idx, z, 128, d_idx, d_slot, tm, coef, cnt, dh, s1, 128, 128, dw, 0, ws, lp, nullptr,
                                         nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr); }),
           2.0 * s1 * 128 * 128, 16.0 * s1 * 128);
  }
  run_variants(x, y, z, tm, dh, w, d_idx, n, s1, bits, ws, dw, report);
  return 0;
}
