// Row-subset GEMMs of the Del operator on the fp32 matrix cores.
//
//   forward / input-gradient :  out[r,:] = act(in[r,:]) @ W (or W^T),  r over an index list
//   weight gradient          :  dW = sum_s a[ia(s),:]^T g[ig(s),:]
//
// Both use v_mfma_f32_32x32x2_f32 (exact fp32 = a k-ordered fmaf chain, 64 FLOP/clk/SIMD: the
// Del GEMM at d=128 has intensity d/4 = 32 flop/B and sits on the compute side of the ridge).
//
// forward kernel: the [d_in, d_out] weight lives in LDS for the life of the block (XOR-swizzled
// columns so the straight fill, the transposed fill and the reads are bank-conflict free); every wave owns
// one 32-row tile at a time and feeds its A operand STRAIGHT from global memory: lane l holds
// row (l&31) and loads one float4 = k-slots {8kb + 4(l>>5) + s, s=0..3}, i.e. the k index of
// MFMA step s is permuted (both operands agree), which turns the gather into 16-byte loads with
// no LDS round trip.  Output rows are written after the k loop, so `in` may alias `out`.
//
// wgrad kernel: the reduction runs over the selected rows; consecutive lanes read consecutive
// features of one row (128-byte coalesced), blocks own contiguous row chunks and write partial
// [d_a, d_b] products that a second kernel reduces in a fixed order (deterministic split-K).
#include "common.h"

namespace gd {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NT>
__global__ __launch_bounds__(256, 2) void rows_gemm_mfma_kernel(
    const float* in, int64_t ld_in, const int32_t* __restrict__ idx, int32_t n_sel,
    const float* __restrict__ w, int32_t d_in, int32_t trans_w, const float* __restrict__ bias, int32_t relu_in,
    float* out, int64_t ld_out, float* __restrict__ save_in) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  constexpr int d_out = 32 * NT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // ---- weight image: wl[k][n] = W[k][n] or W[n][k]
  const int n_w = d_in * d_out;
  // (column index XOR-swizzled with k inside each 32-wide group: conflict-free for the
  //  straight fill, the transposed fill and the B-fragment reads, with no padding)
  if (!trans_w) {
    for (int e = tid; e < n_w; e += 256) {
      const int k = e / d_out, n = e % d_out;
      wl[k * d_out + ((n & ~31) | ((n ^ k) & 31))] = w[e];
    }
  } else {
    for (int e = tid; e < n_w; e += 256) {
      const int k = e % d_in, n = e / d_in;
      wl[k * d_out + ((n & ~31) | ((n ^ k) & 31))] = w[e];
    }
  }
  __syncthreads();

  const int n_tiles = (n_sel + 31) >> 5;
  const int r_lo = lane & 31;       // A row owned by this lane
  const int khalf = lane >> 5;      // which half of the 8-wide k block
  for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
    const int s_a = tile * 32 + r_lo;
    const bool live = s_a < n_sel;
    const int64_t row_a = live ? (idx ? idx[s_a] : s_a) : 0;
    const float4* src = reinterpret_cast<const float4*>(in + row_a * ld_in) + khalf;
    float4* sav = save_in ? reinterpret_cast<float4*>(save_in + (int64_t)s_a * d_in) + khalf : nullptr;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int kblocks = d_in >> 3;
    float4 a_next = live ? src[0] : f4_zero();
    for (int kb = 0; kb < kblocks; ++kb) {
      float4 a4 = a_next;
      if (kb + 1 < kblocks) a_next = live ? src[2 * (kb + 1)] : f4_zero();
      if (sav && live) sav[2 * kb] = a4;
      if (relu_in) {
        a4.x = fmaxf(a4.x, 0.f); a4.y = fmaxf(a4.y, 0.f); a4.z = fmaxf(a4.z, 0.f); a4.w = fmaxf(a4.w, 0.f);
      }
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
      const int k0 = kb * 8 + khalf * 4;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* wk = wl + (k0 + s) * d_out + ((r_lo ^ (k0 + s)) & 31);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float b = wk[t * 32];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b, acc[t], 0, 0, 0);
        }
      }
    }

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2) + 4 * khalf;
      const int s_o = tile * 32 + rr;
      if (s_o >= n_sel) continue;
      const int64_t row_o = idx ? idx[s_o] : s_o;
      float* dst = out + row_o * ld_out + r_lo;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float v = acc[t][r];
        if (bias) v += bias[t * 32 + r_lo];
        dst[t * 32] = v;
      }
    }
  }
}

// Fallback for dimensions the MFMA path does not cover: one wave per selected row, the row is
// held in registers (so in/out may alias), one output column at a time.
__global__ __launch_bounds__(256) void rows_gemm_scalar_kernel(
    const float* in, int64_t ld_in, const int32_t* __restrict__ idx, int32_t n_sel,
    const float* __restrict__ w, int32_t d_in, int32_t d_out, int32_t trans_w, const float* __restrict__ bias,
    int32_t relu_in, float* out, int64_t ld_out, float* __restrict__ save_in) {
  constexpr int kMaxPerLane = 16;  // d_in <= 1024
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n_sel) return;
  const int64_t row = idx ? idx[s] : s;
  float xr[kMaxPerLane];
#pragma unroll
  for (int q = 0; q < kMaxPerLane; ++q) {
    const int k = lane + q * kWave;
    float v = k < d_in ? in[row * ld_in + k] : 0.f;
    if (save_in && k < d_in) save_in[(int64_t)s * d_in + k] = v;
    xr[q] = relu_in ? fmaxf(v, 0.f) : v;
  }
  for (int j = 0; j < d_out; ++j) {
    float p = 0.f;
#pragma unroll
    for (int q = 0; q < kMaxPerLane; ++q) {
      const int k = lane + q * kWave;
      if (k < d_in) p = fmaf(xr[q], trans_w ? w[(int64_t)j * d_in + k] : w[(int64_t)k * d_out + j], p);
    }
    p = wave_sum(p);
    if (lane == 0) out[row * ld_out + j] = p + (bias ? bias[j] : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient
template <int TPW>  // output tiles per wave (tiles = (d_a/32)*(d_b/32) dealt round-robin to 4 waves)
__global__ __launch_bounds__(256, 2) void rows_wgrad_mfma_kernel(
    const float* __restrict__ a, int64_t ld_a, const int32_t* __restrict__ a_idx, const float* __restrict__ g,
    int64_t ld_g, const int32_t* __restrict__ g_idx, const float* __restrict__ relu_mask, int32_t n_sel,
    int32_t d_a, int32_t d_b, int32_t rows_per_block, float* __restrict__ partials) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c_lo = lane & 31;
  const int khalf = lane >> 5;
  const int tb = d_b >> 5;
  const int n_tiles = (d_a >> 5) * tb;

  f32x16 acc[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int s_begin = blockIdx.x * rows_per_block;
  const int s_end = min(n_sel, s_begin + rows_per_block);
  // 8 rows per iteration: lanes < 32 take rows s0..s0+3, lanes >= 32 rows s0+4..s0+7
  for (int s0 = s_begin; s0 < s_end; s0 += 8) {
    int64_t ra[4], rg[4];
    bool ok[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ss = s0 + khalf * 4 + s;
      ok[s] = ss < s_end;
      ra[s] = ok[s] ? (a_idx ? a_idx[ss] : ss) : 0;
      rg[s] = ok[s] ? (g_idx ? g_idx[ss] : ss) : 0;
    }
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int id = wave + 4 * q;
      if (id >= n_tiles) continue;
      const int ca = (id / tb) * 32 + c_lo;
      const int cb = (id % tb) * 32 + c_lo;
      float av[4], gv[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        av[s] = ok[s] ? a[ra[s] * ld_a + ca] : 0.f;
        float gg = ok[s] ? g[rg[s] * ld_g + cb] : 0.f;
        if (relu_mask && ok[s]) gg = relu_mask[rg[s] * ld_g + cb] > 0.f ? gg : 0.f;
        gv[s] = gg;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], gv[s], acc[q], 0, 0, 0);
    }
  }

  float* dst = partials + (int64_t)blockIdx.x * d_a * d_b;
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int id = wave + 4 * q;
    if (id >= n_tiles) continue;
    const int i0 = (id / tb) * 32, j0 = (id % tb) * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2) + 4 * khalf;
      dst[(int64_t)(i0 + rr) * d_b + j0 + c_lo] = acc[q][r];
    }
  }
}

// generic dims: one thread per dW element, loop over the block's rows
__global__ __launch_bounds__(256) void rows_wgrad_scalar_kernel(
    const float* __restrict__ a, int64_t ld_a, const int32_t* __restrict__ a_idx, const float* __restrict__ g,
    int64_t ld_g, const int32_t* __restrict__ g_idx, const float* __restrict__ relu_mask, int32_t n_sel,
    int32_t d_a, int32_t d_b, int32_t rows_per_block, float* __restrict__ partials) {
  const int s_begin = blockIdx.x * rows_per_block;
  const int s_end = min(n_sel, s_begin + rows_per_block);
  float* dst = partials + (int64_t)blockIdx.x * d_a * d_b;
  for (int e = threadIdx.x; e < d_a * d_b; e += 256) {
    const int i = e / d_b, j = e % d_b;
    float p = 0.f;
    for (int s = s_begin; s < s_end; ++s) {
      const int64_t ra = a_idx ? a_idx[s] : s, rg = g_idx ? g_idx[s] : s;
      float gg = g[rg * ld_g + j];
      if (relu_mask) gg = relu_mask[rg * ld_g + j] > 0.f ? gg : 0.f;
      p = fmaf(a[ra * ld_a + i], gg, p);
    }
    dst[e] = p;
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partials, int32_t n_part,
                                                           int32_t n_elem, int32_t accumulate,
                                                           float* __restrict__ dw) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_elem) return;
  float s = accumulate ? dw[e] : 0.f;
  for (int b = 0; b < n_part; ++b) s += partials[(int64_t)b * n_elem + e];
  dw[e] = s;
}

static inline void wgrad_geometry(int32_t n_sel, int* n_blocks, int* rows_per_block) {
  int nb = (n_sel + 63) / 64;
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  int rpb = (n_sel + nb - 1) / nb;
  rpb = (rpb + 7) / 8 * 8;
  if (rpb < 8) rpb = 8;
  *n_blocks = (n_sel + rpb - 1) / rpb;
  if (*n_blocks < 1) *n_blocks = 1;
  *rows_per_block = rpb;
}

}  // namespace gd

extern "C" int gd_rows_gemm_f32(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel, const float* w,
                                int32_t d_in, int32_t d_out, int32_t trans_w, const float* bias, int32_t relu_in,
                                float* out, int64_t ld_out, float* save_in, void* stream) {
  using namespace gd;
  GD_REQUIRE(in && w && out, GD_E_NULL, "gd_rows_gemm_f32: null pointer");
  GD_REQUIRE(n_sel >= 0 && d_in > 0 && d_out > 0 && ld_in >= d_in && ld_out >= d_out, GD_E_DIM,
             "gd_rows_gemm_f32: bad dims n_sel=%d d_in=%d d_out=%d", n_sel, d_in, d_out);
  GD_REQUIRE(in != out || d_in == d_out, GD_E_DIM, "gd_rows_gemm_f32: in-place needs d_in == d_out");
  if (n_sel == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)d_in * d_out * sizeof(float);
  const bool mfma_ok = (d_out % 32 == 0) && d_out <= 128 && (d_in % 8 == 0) && lds <= 64 * 1024 && aligned16(in) &&
                       (ld_in % 4 == 0) && (!save_in || aligned16(save_in));
  if (mfma_ok) {
    const int n_tiles = (n_sel + 31) / 32;
    int grid = (n_tiles + 3) / 4;
    if (grid > 512) grid = 512;
#define GD_RG_CASE(NT)                                                                                            \
  hipLaunchKernelGGL((rows_gemm_mfma_kernel<NT>), dim3(grid), dim3(256), lds, s, in, ld_in, idx, n_sel, w, d_in, \
                     trans_w, bias, relu_in, out, ld_out, save_in)
    switch (d_out / 32) {
      case 1: GD_RG_CASE(1); break;
      case 2: GD_RG_CASE(2); break;
      case 3: GD_RG_CASE(3); break;
      default: GD_RG_CASE(4); break;
    }
#undef GD_RG_CASE
    return launched("rows_gemm_mfma");
  }
  GD_REQUIRE(d_in <= 1024, GD_E_DIM, "gd_rows_gemm_f32: fallback path needs d_in <= 1024 (got %d)", d_in);
  hipLaunchKernelGGL(rows_gemm_scalar_kernel, dim3((n_sel + 3) / 4), dim3(256), 0, s, in, ld_in, idx, n_sel, w, d_in,
                     d_out, trans_w, bias, relu_in, out, ld_out, save_in);
  return launched("rows_gemm_scalar");
}

extern "C" int64_t gd_rows_gemm_wgrad_workspace(int32_t n_sel, int32_t d_a, int32_t d_b) {
  int nb, rpb;
  gd::wgrad_geometry(n_sel, &nb, &rpb);
  return (int64_t)nb * d_a * d_b;
}

extern "C" int gd_rows_gemm_wgrad_f32(const float* a, int64_t ld_a, const int32_t* a_idx, const float* g, int64_t ld_g,
                                      const int32_t* g_idx, const float* relu_mask, int32_t n_sel, int32_t d_a,
                                      int32_t d_b, float* dw, int32_t accumulate, float* partials, void* stream) {
  using namespace gd;
  GD_REQUIRE(dw && partials, GD_E_NULL, "gd_rows_gemm_wgrad_f32: null output");
  GD_REQUIRE(n_sel == 0 || (a && g), GD_E_NULL, "gd_rows_gemm_wgrad_f32: null input");
  GD_REQUIRE(n_sel >= 0 && d_a > 0 && d_b > 0 && ld_a >= d_a && ld_g >= d_b, GD_E_DIM,
             "gd_rows_gemm_wgrad_f32: bad dims");
  hipStream_t s = (hipStream_t)stream;
  const int n_elem = d_a * d_b;
  int nb = 0, rpb = 8;
  if (n_sel > 0) {
    wgrad_geometry(n_sel, &nb, &rpb);
    const int tiles = (d_a / 32) * (d_b / 32);
    const bool mfma_ok = (d_a % 32 == 0) && (d_b % 32 == 0) && tiles <= 16;
    if (mfma_ok) {
#define GD_WG_CASE(TPW)                                                                                           \
  hipLaunchKernelGGL((rows_wgrad_mfma_kernel<TPW>), dim3(nb), dim3(256), 0, s, a, ld_a, a_idx, g, ld_g, g_idx, \
                     relu_mask, n_sel, d_a, d_b, rpb, partials)
      switch ((tiles + 3) / 4) {
        case 1: GD_WG_CASE(1); break;
        case 2: GD_WG_CASE(2); break;
        case 3: GD_WG_CASE(3); break;
        default: GD_WG_CASE(4); break;
      }
#undef GD_WG_CASE
    } else {
      hipLaunchKernelGGL(rows_wgrad_scalar_kernel, dim3(nb), dim3(256), 0, s, a, ld_a, a_idx, g, ld_g, g_idx,
                         relu_mask, n_sel, d_a, d_b, rpb, partials);
    }
    int rc = launched("rows_wgrad");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n_elem + 255) / 256), dim3(256), 0, s, partials, nb, n_elem,
                     accumulate, dw);
  return launched("wgrad_reduce");
}
