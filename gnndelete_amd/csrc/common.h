// Shared helpers for the gfx950 kernels (error reporting, wave-level primitives).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "gnndelete_hip.h"

namespace gd {

constexpr int kWave = 64;  // CDNA wavefront width (hard-coded on purpose: gfx950 only)

char* error_buffer();  // thread-local, defined in api.cpp
int matrix_split();    // 0 / 6 / 9, see gd_set_matrix_split (api.cpp)
int ws_cu_count();     // compute units of the current device (rows_gemm_ws.hip)
// weight-stationary row GEMM (rows_gemm_ws.hip): GD_OK / error when it took the call, 1 when the shape / mode is not covered
int rows_gemm_ws_try(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel, const float* w, int32_t d_in, int32_t d_out,
                     int32_t trans_w, const float* bias, int32_t relu_in, const uint32_t* gate_bits, uint32_t* sign_out, float* out,
                     int64_t ld_out, float* save_in, void* stream, const float* in_alt, const uint8_t* sel, const float* u1,
                     const float* u2, float* o1, float* o2);

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 256, fmt, ap);
  va_end(ap);
  return code;
}

inline int launched(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-(int)e, "%s: %s", what, hipGetErrorString(e));
  return GD_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define GD_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) return ::gd::fail(code, __VA_ARGS__); \
  } while (0)

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_fma(float a, float4 x, float4 acc) {
  acc.x = fmaf(a, x.x, acc.x);
  acc.y = fmaf(a, x.y, acc.y);
  acc.z = fmaf(a, x.z, acc.z);
  acc.w = fmaf(a, x.w, acc.w);
  return acc;
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 f4_shfl_xor(float4 v, int mask) {
  return make_float4(__shfl_xor(v.x, mask), __shfl_xor(v.y, mask), __shfl_xor(v.z, mask), __shfl_xor(v.w, mask));
}
// v[l] + v[l ^ 32] / v[l ^ 16] in every lane without the LDS crossbar: gfx950's v_permlane{32,16}_swap
// exchanges the upper/lower 32 lanes (odd/even 16-lane rows) of two registers in one VALU op
__device__ __forceinline__ float xor32_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_sum(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Sum over the LPR consecutive lanes of a lane group (LPR a power of two), result in every lane of the
// group, on the VALU only: DPP quad permutes / row mirrors inside a 16-lane row, permlane swaps above it.
// Same pairing tree as an xor butterfly (pairs, quads, eights, ...), so the result is bit-identical to it.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int LPR>
__device__ __forceinline__ float lanes_sum(float v) {
  if (LPR >= 2) v = dpp_add<0xB1>(v);     // quad_perm [1,0,3,2]
  if (LPR >= 4) v = dpp_add<0x4E>(v);     // quad_perm [2,3,0,1]
  if (LPR >= 8) v = dpp_add<0x141>(v);    // row_half_mirror: the other quad of the 8
  if (LPR >= 16) v = dpp_add<0x140>(v);   // row_mirror: the other half of the 16
  if (LPR >= 32) v = xor16_sum(v);
  if (LPR >= 64) v = xor32_sum(v);
  return v;
}

// acc[l] += acc[l ^ off] for off = LPR, 2 LPR, ... < 64 (the sum over the 64 / LPR lane groups)
template <int LPR>
__device__ __forceinline__ float4 f4_group_sum(float4 a) {
#pragma unroll
  for (int off = LPR; off < 8; off <<= 1) a = f4_add(a, f4_shfl_xor(a, off));
  // l ^ 8 inside a 16-lane row is a rotation by 8: one DPP add per float instead of a trip through the LDS crossbar
  if (LPR <= 8) a = make_float4(dpp_add<0x128>(a.x), dpp_add<0x128>(a.y), dpp_add<0x128>(a.z), dpp_add<0x128>(a.w));
  if (LPR <= 16) a = make_float4(xor16_sum(a.x), xor16_sum(a.y), xor16_sum(a.z), xor16_sum(a.w));
  if (LPR <= 32) a = make_float4(xor32_sum(a.x), xor32_sum(a.y), xor32_sum(a.z), xor32_sum(a.w));
  return a;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// torch.optim.Adam (single-tensor CPU path: lerp_, mul_/addcmul_, sqrt/div/add_, addcdiv_) with the
// rounding sequence of that path spelled out, so every kernel that applies it (stand-alone or in a
// reduction epilogue) is bit-identical to the others: hyper-parameters arrive as the Python doubles the
// optimizer holds and are derived in fp64 before one rounding to fp32, as torch's scalar handling does.
struct AdamScalars { float w1, b2, c2, neg_step, rs, eps; };
__device__ __forceinline__ AdamScalars adam_scalars(double lr, double beta1, double beta2, double eps, int t) {
  const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
  AdamScalars s;
  s.w1 = (float)(1.0 - beta1);
  s.b2 = (float)beta2;
  s.c2 = (float)(1.0 - beta2);
  s.neg_step = (float)(-(lr / bc1));
  s.rs = (float)sqrt(bc2);
  s.eps = (float)eps;
  return s;
}
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, const AdamScalars& s) {
#pragma clang fp contract(off)
  m = __builtin_fmaf(g - m, s.w1, m);                       // exp_avg.lerp_(grad, 1 - beta1)
  v = __builtin_fmaf(s.c2 * g, g, v * s.b2);                // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
  const float den = sqrtf(v) / s.rs + s.eps;           // (sqrt / sqrt(bc2)).add_(eps)
  p = p + (s.neg_step * m) / den;                           // param.addcdiv_(exp_avg, denom, -step_size)
}

// Bijective remap of the hardware block id so that the blocks resident on one XCD (b % 8) cover a
// contiguous range of logical block ids (placement is a speed assumption only, never correctness).
__device__ __forceinline__ int xcd_contiguous_block(int b, int n_blocks) {
  constexpr int kXcd = 8;
  const int q = n_blocks / kXcd, r = n_blocks % kXcd;
  const int xcd = b % kXcd, local = b / kXcd;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// lanes-per-row geometry shared by the row-vector kernels: a row of d floats is d/4 float4
// vectors; LPR lanes (power of two <= 64) cooperate on one row, 64/LPR rows per wave.
inline int lanes_per_row(int d4) {
  int l = 1;
  while (l < d4 && l < 64) l <<= 1;
  return l;
}

}  // namespace gd
