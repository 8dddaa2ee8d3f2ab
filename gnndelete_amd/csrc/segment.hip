// Segment softmax over CSR rows and per-pair row dot products: the attention of the reference's RGATConv
// (framework/models/rgat.py:188-241: additive attention logits per edge, `softmax(alpha, index, ptr, size_i)` across the
// relations of a target node, :322-337 messages alpha_e x_j W_r) as kernels instead of index_reduce / index_add chains.
//
//   segment_softmax fwd   alpha[k] = exp(e[k] - max_row) / (sum_row exp(e - max_row) + 1e-16)       (PyG utils.softmax)
//                   bwd   de[k] = alpha[k] (dalpha[k] - sum_row alpha dalpha)
//   rowpair_dot           out[k] = < a[ia[k], :], b[ib[k], :] >      (d alpha_e = < dm[run(e)], x[src(e)] >)
//
// One wave per row, lanes stride over the row's entries, wave reductions in a fixed order (bit-reproducible); rows are
// short (a node's in-edges), the kernels are latency-bound streams over the edge arrays.
#include "common.h"

namespace gd {

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

__global__ __launch_bounds__(256) void segment_softmax_fwd_kernel(const int32_t* __restrict__ rowptr, const float* __restrict__ e,
                                                                  int32_t n_rows, float* __restrict__ alpha) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int s = rowptr[row], t = rowptr[row + 1];
  if (s == t) return;
  float m = -INFINITY;
  for (int k = s + lane; k < t; k += kWave) m = fmaxf(m, e[k]);
  m = wave_max_f(m);
  float sum = 0.f;
  for (int k = s + lane; k < t; k += kWave) sum += __expf(e[k] - m);
  sum = wave_sum(sum) + 1e-16f;
  for (int k = s + lane; k < t; k += kWave) alpha[k] = __expf(e[k] - m) / sum;
}

__global__ __launch_bounds__(256) void segment_softmax_bwd_kernel(const int32_t* __restrict__ rowptr, const float* __restrict__ alpha,
                                                                  const float* __restrict__ dalpha, int32_t n_rows,
                                                                  float* __restrict__ de) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int s = rowptr[row], t = rowptr[row + 1];
  if (s == t) return;
  float dot = 0.f;
  for (int k = s + lane; k < t; k += kWave) dot = fmaf(alpha[k], dalpha[k], dot);
  dot = wave_sum(dot);
  for (int k = s + lane; k < t; k += kWave) de[k] = alpha[k] * (dalpha[k] - dot);
}

// LPR lanes x float4 per pair (d4 <= LPR), 64 / LPR pairs per wave
template <int LPR>
__global__ __launch_bounds__(256) void rowpair_dot_kernel(const float* __restrict__ a, int64_t ld_a, const int32_t* __restrict__ ia,
                                                          const float* __restrict__ b, int64_t ld_b, const int32_t* __restrict__ ib,
                                                          int64_t n, int32_t d4, float* __restrict__ out) {
  constexpr int G = kWave / LPR;
  const int lane = threadIdx.x & 63, g = lane / LPR, li = lane % LPR;
  const int64_t k = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * G + g;
  float acc = 0.f;
  if (k < n && li < d4) {
    const float4 va = reinterpret_cast<const float4*>(a + (int64_t)ia[k] * ld_a)[li];
    const float4 vb = reinterpret_cast<const float4*>(b + (int64_t)ib[k] * ld_b)[li];
    acc = va.x * vb.x;
    acc = fmaf(va.y, vb.y, acc); acc = fmaf(va.z, vb.z, acc); acc = fmaf(va.w, vb.w, acc);
  }
  acc = lanes_sum<LPR>(acc);
  if (k < n && li == 0) out[k] = acc;
}

__global__ __launch_bounds__(256) void rowpair_dot_scalar_kernel(const float* __restrict__ a, int64_t ld_a, const int32_t* __restrict__ ia,
                                                                 const float* __restrict__ b, int64_t ld_b, const int32_t* __restrict__ ib,
                                                                 int64_t n, int32_t d, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n) return;
  float acc = 0.f;
  for (int c = lane; c < d; c += kWave) acc = fmaf(a[(int64_t)ia[k] * ld_a + c], b[(int64_t)ib[k] * ld_b + c], acc);
  acc = wave_sum(acc);
  if (lane == 0) out[k] = acc;
}

}  // namespace gd

extern "C" int gd_segment_softmax_f32(const int32_t* rowptr, const float* e, int32_t n_rows, float* alpha, void* stream) {
  GD_REQUIRE(rowptr && (e || n_rows == 0) && (alpha || n_rows == 0), GD_E_NULL, "gd_segment_softmax_f32: null pointer");
  GD_REQUIRE(n_rows >= 0, GD_E_DIM, "gd_segment_softmax_f32: n_rows < 0");
  if (n_rows == 0) return GD_OK;
  hipLaunchKernelGGL(gd::segment_softmax_fwd_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, rowptr, e, n_rows, alpha);
  return gd::launched("segment_softmax_fwd");
}

extern "C" int gd_segment_softmax_bwd_f32(const int32_t* rowptr, const float* alpha, const float* dalpha, int32_t n_rows,
                                          float* de, void* stream) {
  GD_REQUIRE(rowptr && ((alpha && dalpha && de) || n_rows == 0), GD_E_NULL, "gd_segment_softmax_bwd_f32: null pointer");
  GD_REQUIRE(n_rows >= 0, GD_E_DIM, "gd_segment_softmax_bwd_f32: n_rows < 0");
  if (n_rows == 0) return GD_OK;
  hipLaunchKernelGGL(gd::segment_softmax_bwd_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, rowptr, alpha, dalpha,
                     n_rows, de);
  return gd::launched("segment_softmax_bwd");
}

extern "C" int gd_rowpair_dot_f32(const float* a, int64_t ld_a, const int32_t* ia, const float* b, int64_t ld_b, const int32_t* ib,
                                  int64_t n, int32_t d, float* out, void* stream) {
  using namespace gd;
  GD_REQUIRE(n == 0 || (a && b && ia && ib && out), GD_E_NULL, "gd_rowpair_dot_f32: null pointer");
  GD_REQUIRE(n >= 0 && d > 0 && ld_a >= d && ld_b >= d, GD_E_DIM, "gd_rowpair_dot_f32: bad dims");
  if (n == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = d % 4 == 0 && d <= 256 && ld_a % 4 == 0 && ld_b % 4 == 0 && aligned16(a) && aligned16(b);
  if (!vec) {
    hipLaunchKernelGGL(rowpair_dot_scalar_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, a, ld_a, ia, b, ld_b, ib, n, d, out);
    return launched("rowpair_dot_scalar");
  }
  const int d4 = d / 4;
#define GD_RPD(LPR)                                                                                                     \
  hipLaunchKernelGGL((rowpair_dot_kernel<LPR>), dim3((unsigned)((n + 4 * (64 / LPR) - 1) / (4 * (64 / LPR)))), dim3(256), 0, s, a, \
                     ld_a, ia, b, ld_b, ib, n, d4, out)
  if (d4 <= 4) GD_RPD(4);
  else if (d4 <= 8) GD_RPD(8);
  else if (d4 <= 16) GD_RPD(16);
  else if (d4 <= 32) GD_RPD(32);
  else GD_RPD(64);
#undef GD_RPD
  return launched("rowpair_dot");
}
