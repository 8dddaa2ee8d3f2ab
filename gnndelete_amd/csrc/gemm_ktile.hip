// Dense transform with an ARBITRARILY WIDE reduction dimension on the fp32 matrix cores:
//
//     out[r, :] = in[r, :] @ W (+ bias),   in [M, K] (K up to the bag-of-words widths: 1,639 / 8,710), W [K, N], N <= 128
//
// - the layer-1 transform x W1^T of the citation graphs (framework/models/gcn.py:11-12 through PyG's Linear), which
// the whole-weight-in-LDS row kernel (rows_gemm.hip) cannot take: its image is d_in * d_out * 4 bytes.
//
// Mapping to CDNA4: a block of 8 waves owns 256 rows (one 32-row tile per wave, operand rows straight from global
// memory, 64 contiguous bytes per lane and 32-wide k chunk, exactly as in rows_gemm.hip) and walks a k RANGE; the
// weight is streamed through LDS in 32-row chunks (16 KB at N = 128, double buffered, one barrier per chunk), in
// the tile-interleaved layout whose NT operands per k step are one 16-byte read.  M = 17,716 rows is only 554
// wave tiles for 1,024 SIMDs, so the k dimension is split over up to 8 blocks per row group (grid.y) whose
// partial products a second kernel adds in split order - deterministic, no atomics.
// v_mfma_f32_32x32x2_f32: exact fp32 (a k-ordered fmaf chain per split).
#include "common.h"

namespace gd {

using f32x16k = __attribute__((ext_vector_type(16))) float;
constexpr int kKtThreads = 512;

template <int NT>
__global__ __launch_bounds__(kKtThreads, 4) void gemm_ktile_mfma_kernel(
    const float* __restrict__ in, int64_t ld_in, const int32_t* __restrict__ idx, int32_t n_rows,
    const float* __restrict__ w, int32_t n_chunks, int32_t chunks_per_split, const float* __restrict__ bias,
    float* __restrict__ out, int64_t ld_out, int64_t split_stride, int32_t scatter) {
  constexpr int NTP = NT == 3 ? 4 : NT;
  constexpr int N = 32 * NT;
  constexpr int kChunkFloats = 32 * 32 * NTP;
  constexpr int PAIRS = (32 * 32) / kKtThreads;          // (k, r) pairs of a chunk per thread
  __shared__ __attribute__((aligned(16))) float wl[2][kChunkFloats];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r_lo = lane & 31, khalf = lane >> 5;
  const int c0 = blockIdx.y * chunks_per_split, c1 = min(n_chunks, c0 + chunks_per_split);
  const int tile = blockIdx.x * 8 + wave;
  const int n_tiles = (n_rows + 31) >> 5;
  const bool has_tile = tile < n_tiles;
  const int s_a = min(tile * 32 + r_lo, n_rows - 1);
  const int64_t row = idx ? idx[has_tile ? s_a : 0] : (has_tile ? s_a : 0);
  const float4* src = reinterpret_cast<const float4*>(in + row * ld_in) + khalf * 4;

  float wreg[PAIRS][NTP];
  auto load_w = [&](int c) {
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      const int e = tid + p * kKtThreads, kk = e >> 5, r = e & 31;
      const float* wp = w + (int64_t)(c * 32 + kk) * N + r;
#pragma unroll
      for (int t = 0; t < NTP; ++t) wreg[p][t] = t < NT ? wp[32 * t] : 0.f;
    }
  };
  auto stash_w = [&](int buf) {
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      float* dst = wl[buf] + (tid + p * kKtThreads) * NTP;
      if (NTP == 4) *reinterpret_cast<float4*>(dst) = make_float4(wreg[p][0], wreg[p][1 % NTP], wreg[p][2 % NTP], wreg[p][3 % NTP]);
      else if (NTP == 2) *reinterpret_cast<float2*>(dst) = make_float2(wreg[p][0], wreg[p][1 % NTP]);
      else dst[0] = wreg[p][0];
    }
  };

  f32x16k acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  float4 a_next[4];
  if (c0 < c1) {
    load_w(c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) a_next[i] = src[c0 * 8 + i];
    stash_w(0);
  }
  __syncthreads();
  int cur = 0;
  for (int c = c0; c < c1; ++c) {
    float4 a4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a4[i] = a_next[i];
    const bool more = c + 1 < c1;
    if (more) {                                          // next chunk's operands in flight behind this chunk's MFMAs
#pragma unroll
      for (int i = 0; i < 4; ++i) a_next[i] = src[(c + 1) * 8 + i];
      load_w(c + 1);
    }
    const float* wk = wl[cur] + (khalf * 16 * 32 + r_lo) * NTP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float av[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* wp = wk + (i * 4 + s) * 32 * NTP;
        float wv[NTP];
        if (NTP == 4) {
          const float4 f = *reinterpret_cast<const float4*>(wp);
          wv[0] = f.x; wv[1 % NTP] = f.y; wv[2 % NTP] = f.z; wv[3 % NTP] = f.w;
        } else if (NTP == 2) {
          const float2 f = *reinterpret_cast<const float2*>(wp);
          wv[0] = f.x; wv[1 % NTP] = f.y;
        } else {
          wv[0] = wp[0];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t], av[s], acc[t], 0, 0, 0);
      }
    }
    if (more) stash_w(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  if (!has_tile || tile * 32 + r_lo >= n_rows) return;
  // D[i][j]: j = lane & 31 = sample, feature 32 t + 8 q + 4 khalf + c in acc[t][4 q + c] (as rows_gemm.hip)
  const int64_t orow = (scatter && idx) ? row : s_a;
  float* dst = out + (int64_t)blockIdx.y * split_stride + orow * ld_out + 4 * khalf;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
      if (bias) v = f4_add(v, *reinterpret_cast<const float4*>(bias + 32 * t + 8 * q + 4 * khalf));
      *reinterpret_cast<float4*>(dst + 32 * t + 8 * q) = v;
    }
}

// out[row(s), :] = bias + sum over the k splits (in split order) of partial[split][s, :]
__global__ __launch_bounds__(256) void gemm_ktile_reduce_kernel(const float* __restrict__ partial, int64_t split_stride,
                                                                int32_t n_splits, const int32_t* __restrict__ idx,
                                                                int32_t n_rows, int32_t n4, const float* __restrict__ bias,
                                                                float* __restrict__ out, int64_t ld_out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)n_rows * n4) return;
  const int s = (int)(e / n4), v = (int)(e % n4);
  float4 acc = bias ? reinterpret_cast<const float4*>(bias)[v] : f4_zero();
  const float4* p = reinterpret_cast<const float4*>(partial) + e;
  int k = 0;
  for (; k + 2 <= n_splits; k += 2) {
    const float4 a = p[(int64_t)k * (split_stride / 4)], b = p[(int64_t)(k + 1) * (split_stride / 4)];
    acc = f4_add(f4_add(acc, a), b);
  }
  if (k < n_splits) acc = f4_add(acc, p[(int64_t)k * (split_stride / 4)]);
  const int64_t row = idx ? idx[s] : s;
  reinterpret_cast<float4*>(out + row * ld_out)[v] = acc;
}

static void ktile_geometry(int32_t n_rows, int32_t k, int* n_groups, int* n_splits, int* cps) {
  const int groups = (n_rows + 255) / 256, chunks = k / 32;
  int want = groups > 0 ? 400 / groups : 1;
  if (want > 8) want = 8;
  if (want > chunks / 4) want = chunks / 4;
  if (want < 1) want = 1;
  const int per = (chunks + want - 1) / want;
  *n_groups = groups;
  *cps = per;
  *n_splits = (chunks + per - 1) / per;
}

}  // namespace gd

extern "C" int64_t gd_gemm_f32_workspace(int32_t n_rows, int32_t k, int32_t n) {
  if (n_rows <= 0 || k <= 0 || k % 32) return 0;
  int g, s, c;
  gd::ktile_geometry(n_rows, k, &g, &s, &c);
  return s > 1 ? (int64_t)s * n_rows * n : 0;
}

extern "C" int gd_gemm_f32(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_rows, const float* w, int32_t k,
                           int32_t n, const float* bias, float* out, int64_t ld_out, float* workspace, void* stream) {
  using namespace gd;
  GD_REQUIRE(in && w && out, GD_E_NULL, "gd_gemm_f32: null pointer");
  GD_REQUIRE(n_rows >= 0 && k > 0 && k % 32 == 0 && n % 32 == 0 && n >= 32 && n <= 128 && ld_in >= k && ld_out >= n &&
                 ld_in % 4 == 0 && ld_out % 4 == 0, GD_E_DIM,
             "gd_gemm_f32: needs k %% 32 == 0 (pad with zeros), n in {32, 64, 96, 128}, 16-byte row pitches (k=%d n=%d)", k, n);
  GD_REQUIRE(aligned16(in) && aligned16(out) && aligned16(w) && (!bias || aligned16(bias)) && in != out, GD_E_ALIGN,
             "gd_gemm_f32: unaligned or aliasing pointer");
  if (n_rows == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  int groups, splits, cps;
  ktile_geometry(n_rows, k, &groups, &splits, &cps);
  GD_REQUIRE(splits == 1 || (workspace && aligned16(workspace)), GD_E_NULL, "gd_gemm_f32: workspace of gd_gemm_f32_workspace() floats");
  const dim3 grid(groups, splits), block(kKtThreads);
  const int n_chunks = k / 32;
  float* dst = splits == 1 ? out : workspace;
  const int64_t ld_dst = splits == 1 ? ld_out : n;
  const int64_t stride = splits == 1 ? 0 : (int64_t)n_rows * n;
  const float* b = splits == 1 ? bias : nullptr;
#define GD_KT_CASE(NT) \
  hipLaunchKernelGGL((gemm_ktile_mfma_kernel<NT>), grid, block, 0, s, in, ld_in, idx, n_rows, w, n_chunks, cps, b, dst, ld_dst, stride, splits == 1)
  switch (n / 32) {
    case 1: GD_KT_CASE(1); break;
    case 2: GD_KT_CASE(2); break;
    case 3: GD_KT_CASE(3); break;
    default: GD_KT_CASE(4); break;
  }
#undef GD_KT_CASE
  int rc = launched("gemm_ktile");
  if (rc || splits == 1) return rc;
  const int n4 = n / 4;
  const int64_t total = (int64_t)n_rows * n4;
  hipLaunchKernelGGL(gemm_ktile_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, workspace, stride, splits, idx,
                     n_rows, n4, bias, out, ld_out);
  return launched("gemm_ktile_reduce");
}
