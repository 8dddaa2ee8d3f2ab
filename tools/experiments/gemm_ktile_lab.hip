// K-tiled MFMA GEMM (csrc/gemm_ktile.hip) vs rocBLAS sgemm at the wide-input shapes of BASELINE's configs 1 / 2.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -I../../gnndelete_amd/csrc gemm_ktile_lab.hip -lrocblas
#include "../../gnndelete_amd/csrc/gemm_ktile.hip"
#include <rocblas/rocblas.h>
#include <vector>
namespace gd { char* error_buffer() { static thread_local char b[256]; return b; } }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <typename F> static double time_us(F launch, int reps = 10) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3;
}
__global__ void fill_kernel(float* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t a = (uint32_t)i * 2654435761u + seed; a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15;
    p[i] = (float)(a & 0xffff) * (1.0f / 65536.0f) - 0.5f;
  }
}
int main() {
  rocblas_handle h; rocblas_create_handle(&h);
  const int shapes[][3] = {{17716, 1664, 128}, {19793, 8736, 128}, {235868, 128, 128}, {17716, 1664, 64}};
  for (auto& sh : shapes) {
    const int M = sh[0], K = sh[1], N = sh[2];
    float *x, *w, *y, *y2, *ws;
    CK(hipMalloc(&x, (size_t)M * K * 4)); CK(hipMalloc(&w, (size_t)K * N * 4)); CK(hipMalloc(&y, (size_t)M * N * 4)); CK(hipMalloc(&y2, (size_t)M * N * 4));
    const int64_t wsn = gd_gemm_f32_workspace(M, K, N);
    CK(hipMalloc(&ws, (size_t)std::max<int64_t>(wsn, 4) * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, x, (size_t)M * K, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, w, (size_t)K * N, 2u);
    CK(hipDeviceSynchronize());
    const double us = time_us([&] { int rc = gd_gemm_f32(x, K, nullptr, M, w, K, N, nullptr, y, N, ws, nullptr); if (rc) { printf("rc %d %s\n", rc, gd::error_buffer()); exit(1); } });
    // rocBLAS: row-major y[M,N] = x[M,K] w[K,N]  ==  column-major y^T[N,M] = w^T[N,K] x^T[K,M]
    const float one = 1.f, zero = 0.f;
    const double usr = time_us([&] { rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, N, M, K, &one, w, N, x, K, &zero, y2, N); });
    std::vector<float> a((size_t)M * N), b((size_t)M * N);
    CK(hipMemcpy(a.data(), y, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), y2, b.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0; for (size_t i = 0; i < a.size(); ++i) { num += (double)(a[i] - b[i]) * (a[i] - b[i]); den += (double)b[i] * b[i]; }
    const double fl = 2.0 * M * K * N;
    printf("M=%d K=%d N=%d: gd_gemm_f32 %.1f us = %.1f TF (workspace %.1f MB)   rocBLAS %.1f us = %.1f TF   rel diff %.2e\n", M, K, N, us,
           fl / us / 1e6, wsn * 4 / 1e6, usr, fl / usr / 1e6, sqrt(num / den));
    hipFree(x); hipFree(w); hipFree(y); hipFree(y2); hipFree(ws);
  }
  return 0;
}
