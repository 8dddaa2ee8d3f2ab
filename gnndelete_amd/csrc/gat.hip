// GAT (heads = 1) attention-scored aggregation: fused edge-score / segment-softmax / weighted
// gather in one kernel (the reference runs ~8 separate gather / scatter kernels per layer), and
// its backward as two kernels (target-major pass for the softmax Jacobian, source-major pass for
// the message gradient).  One wave per CSR row, LPR lanes x float4 per feature row, G = 64/LPR
// neighbours in flight; scores of up to 64 in-edges live in one register per lane.
#include <math.h>

#include "common.h"

namespace gd {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float leaky(float s, float slope) { return s > 0.f ? s : slope * s; }

template <int LPR, int VPL>
__global__ __launch_bounds__(256) void gat_fwd_kernel(const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ col,
                                                      const float* __restrict__ a_src, const float* __restrict__ a_dst,
                                                      const float* __restrict__ h, int64_t ldh, float* __restrict__ y,
                                                      int64_t ldy, const float* __restrict__ bias,
                                                      float* __restrict__ alpha_out, float slope, int32_t n_rows,
                                                      int32_t d4) {
  constexpr int G = kWave / LPR;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int g = lane / LPR, li = lane % LPR;
  const int start = rowptr[row], end = rowptr[row + 1];
  const float ad = a_dst[row];

  // pass 1: segment max
  float mx = -INFINITY;
  for (int k = start + lane; k < end; k += kWave) mx = fmaxf(mx, leaky(a_src[col[k]] + ad, slope));
  mx = wave_max(mx);
  // pass 2: segment sum of exp
  float sm = 0.f;
  for (int k = start + lane; k < end; k += kWave) sm += expf(leaky(a_src[col[k]] + ad, slope) - mx);
  sm = wave_sum(sm);
  const float inv = 1.0f / (sm + 1e-16f);

  // pass 3: weighted gather
  float4 acc[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = f4_zero();
  for (int base = start; base < end; base += kWave) {
    const int k = base + lane;
    const bool live = k < end;
    const int c = live ? col[k] : 0;
    const float w = live ? expf(leaky(a_src[c] + ad, slope) - mx) * inv : 0.f;
    if (alpha_out && live) alpha_out[k] = w;
    const int cnt = min(kWave, end - base);
    const int trips = (cnt + G - 1) / G;

    for (int it = 0; it < trips; ++it) {
      const int j = it * G + g;
      const int cj = __shfl(c, j);
      const float wj = __shfl(w, j);
      const float4* hr = reinterpret_cast<const float4*>(h + (int64_t)cj * ldh);
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int vec = li + v * LPR;
        if (vec < d4) acc[v] = f4_fma(wj, hr[vec], acc[v]);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VPL; ++v)
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) acc[v] = f4_add(acc[v], f4_shfl_xor(acc[v], off));
  if (g != 0) return;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int vec = li + v * LPR;
    if (vec >= d4) continue;
    float4 o = acc[v];
    if (bias) o = f4_add(o, reinterpret_cast<const float4*>(bias)[vec]);
    reinterpret_cast<float4*>(y + (int64_t)row * ldy)[vec] = o;
  }
}

// backward, target-major: d_alpha_e = <dy_i, h_j>; de_e = alpha_e (d_alpha_e - sum alpha d_alpha) * leaky'
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void gat_bwd_row_kernel(const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col,
                                                          const float* __restrict__ alpha,
                                                          const float* __restrict__ a_src,
                                                          const float* __restrict__ a_dst, const float* __restrict__ h,
                                                          int64_t ldh, const float* __restrict__ dy, int64_t lddy,
                                                          float* __restrict__ de, float* __restrict__ da_dst,
                                                          float slope, int32_t n_rows, int32_t d4) {
  constexpr int G = kWave / LPR;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int g = lane / LPR, li = lane % LPR;
  const int start = rowptr[row], end = rowptr[row + 1];
  float4 dyr[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int vec = li + v * LPR;
    dyr[v] = vec < d4 ? reinterpret_cast<const float4*>(dy + (int64_t)row * lddy)[vec] : f4_zero();
  }
  // pass A: d_alpha per edge (stored in de), t = sum alpha * d_alpha
  float t = 0.f;
  for (int base = start; base < end; base += G) {
    const int k = base + g;
    float p = 0.f;
    if (k < end) {
      const float4* hr = reinterpret_cast<const float4*>(h + (int64_t)col[k] * ldh);
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int vec = li + v * LPR;
        if (vec >= d4) continue;
        const float4 hv = hr[vec];
        p = fmaf(dyr[v].x, hv.x, p); p = fmaf(dyr[v].y, hv.y, p); p = fmaf(dyr[v].z, hv.z, p); p = fmaf(dyr[v].w, hv.w, p);
      }
    }
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) p += __shfl_xor(p, off);
    if (k < end && li == 0) {
      de[k] = p;
      t = fmaf(alpha[k], p, t);
    }
  }
  t = wave_sum(t);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // pass-A stores of other lanes -> pass-B loads
  __builtin_amdgcn_wave_barrier();
  // pass B
  const float ad = a_dst[row];
  float sd = 0.f;
  for (int k = start + lane; k < end; k += kWave) {
    const float s = a_src[col[k]] + ad;
    const float v = alpha[k] * (de[k] - t) * (s > 0.f ? 1.0f : slope);
    de[k] = v;
    sd += v;
  }
  sd = wave_sum(sd);
  if (lane == 0) da_dst[row] = sd;
}

// backward, source-major: dh_j = sum_{j->i} alpha_e dy_i ; da_src[j] = sum de_e
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void gat_bwd_col_kernel(const int32_t* __restrict__ rowptr_t,
                                                          const int32_t* __restrict__ col_t,
                                                          const int32_t* __restrict__ perm_t,
                                                          const float* __restrict__ alpha, const float* __restrict__ de,
                                                          const float* __restrict__ dy, int64_t lddy,
                                                          float* __restrict__ dh, int64_t lddh,
                                                          float* __restrict__ da_src, int32_t n_rows, int32_t d4) {
  constexpr int G = kWave / LPR;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int g = lane / LPR, li = lane % LPR;
  const int start = rowptr_t[row], end = rowptr_t[row + 1];
  float4 acc[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = f4_zero();
  float sa = 0.f;
  for (int base = start; base < end; base += kWave) {
    const int k = base + lane;
    const bool live = k < end;
    const int c = live ? col_t[k] : 0;
    const int p = live ? perm_t[k] : 0;
    const float w = live ? alpha[p] : 0.f;
    if (live) sa += de[p];
    const int cnt = min(kWave, end - base);
    const int trips = (cnt + G - 1) / G;

    for (int it = 0; it < trips; ++it) {
      const int j = it * G + g;
      const int cj = __shfl(c, j);
      const float wj = __shfl(w, j);
      const float4* dr = reinterpret_cast<const float4*>(dy + (int64_t)cj * lddy);
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int vec = li + v * LPR;
        if (vec < d4) acc[v] = f4_fma(wj, dr[vec], acc[v]);
      }
    }
  }
  sa = wave_sum(sa);
  if (lane == 0) da_src[row] = sa;
#pragma unroll
  for (int v = 0; v < VPL; ++v)
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) acc[v] = f4_add(acc[v], f4_shfl_xor(acc[v], off));
  if (g != 0) return;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int vec = li + v * LPR;
    if (vec < d4) reinterpret_cast<float4*>(dh + (int64_t)row * lddh)[vec] = acc[v];
  }
}

#define GD_GAT_DISPATCH(KERNEL, ...)                                                         \
  do {                                                                                       \
    const int lpr = lanes_per_row(d4);                                                       \
    switch (lpr) {                                                                           \
      case 1: hipLaunchKernelGGL((KERNEL<1, 1>), grid, dim3(256), 0, s, __VA_ARGS__); break;  \
      case 2: hipLaunchKernelGGL((KERNEL<2, 1>), grid, dim3(256), 0, s, __VA_ARGS__); break;  \
      case 4: hipLaunchKernelGGL((KERNEL<4, 1>), grid, dim3(256), 0, s, __VA_ARGS__); break;  \
      case 8: hipLaunchKernelGGL((KERNEL<8, 1>), grid, dim3(256), 0, s, __VA_ARGS__); break;  \
      case 16: hipLaunchKernelGGL((KERNEL<16, 1>), grid, dim3(256), 0, s, __VA_ARGS__); break; \
      case 32: hipLaunchKernelGGL((KERNEL<32, 1>), grid, dim3(256), 0, s, __VA_ARGS__); break; \
      default:                                                                               \
        if (d4 <= 64) hipLaunchKernelGGL((KERNEL<64, 1>), grid, dim3(256), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<64, 4>), grid, dim3(256), 0, s, __VA_ARGS__);          \
    }                                                                                        \
  } while (0)

}  // namespace gd

extern "C" int gd_gat_aggregate_f32(const int32_t* rowptr, const int32_t* col, const float* a_src, const float* a_dst,
                                    const float* h, int64_t ldh, float* y, int64_t ldy, const float* bias,
                                    float* alpha_out, float slope, int32_t n_rows, int32_t d, void* stream) {
  using namespace gd;
  GD_REQUIRE(rowptr && col && a_src && a_dst && h && y, GD_E_NULL, "gd_gat_aggregate_f32: null pointer");
  GD_REQUIRE(n_rows >= 0 && d > 0 && d % 4 == 0 && d <= 1024 && ldh % 4 == 0 && ldy % 4 == 0, GD_E_DIM,
             "gd_gat_aggregate_f32: d=%d must be a multiple of 4 (<=1024) with 16-byte row strides", d);
  GD_REQUIRE(aligned16(h) && aligned16(y) && (!bias || aligned16(bias)), GD_E_ALIGN, "gd_gat_aggregate_f32: unaligned");
  if (n_rows == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((n_rows + 3) / 4);
  const int d4 = d / 4;
  GD_GAT_DISPATCH(gat_fwd_kernel, rowptr, col, a_src, a_dst, h, ldh, y, ldy, bias, alpha_out, slope, n_rows, d4);
  return launched("gat_fwd");
}

extern "C" int gd_gat_aggregate_bwd_f32(const int32_t* rowptr, const int32_t* col, const float* alpha,
                                        const int32_t* rowptr_t, const int32_t* col_t, const int32_t* perm_t,
                                        const float* a_src, const float* a_dst, const float* h, int64_t ldh,
                                        const float* dy, int64_t lddy, float* dh, int64_t lddh, float* da_src,
                                        float* da_dst, float* de, float slope, int32_t n_rows, int32_t d,
                                        void* stream) {
  using namespace gd;
  GD_REQUIRE(rowptr && col && alpha && rowptr_t && col_t && perm_t && a_src && a_dst && h && dy && dh && da_src &&
                 da_dst && de, GD_E_NULL, "gd_gat_aggregate_bwd_f32: null pointer");
  GD_REQUIRE(n_rows >= 0 && d > 0 && d % 4 == 0 && d <= 1024 && ldh % 4 == 0 && lddy % 4 == 0 && lddh % 4 == 0,
             GD_E_DIM, "gd_gat_aggregate_bwd_f32: bad dims");
  GD_REQUIRE(aligned16(h) && aligned16(dy) && aligned16(dh), GD_E_ALIGN, "gd_gat_aggregate_bwd_f32: unaligned");
  if (n_rows == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((n_rows + 3) / 4);
  const int d4 = d / 4;
  GD_GAT_DISPATCH(gat_bwd_row_kernel, rowptr, col, alpha, a_src, a_dst, h, ldh, dy, lddy, de, da_dst, slope, n_rows, d4);
  int rc = launched("gat_bwd_row");
  if (rc) return rc;
  GD_GAT_DISPATCH(gat_bwd_col_kernel, rowptr_t, col_t, perm_t, alpha, de, dy, lddy, dh, lddh, da_src, n_rows, d4);
  return launched("gat_bwd_col");
}
