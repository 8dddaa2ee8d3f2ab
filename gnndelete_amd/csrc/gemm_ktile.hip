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
// the tile-interleaved layout whose NT operands per k step are one 16-byte read.
// v_mfma_f32_32x32x2_f32: exact fp32 (a k-ordered fmaf chain per piece).
#include "common.h"

namespace gd {

using f32x16k = __attribute__((ext_vector_type(16))) float;
constexpr int kKtThreads = 512;

// Work partition ("stream-K"): the (row group, k chunk) units are numbered group-major and cut into equal ranges of
// `per` units, one range per block - a block walks its range and, whenever the range leaves a row group, parks the
// accumulators as one PIECE of that group.  Equal ranges instead of a groups x splits grid because M = 17,716 is 70 row
// groups: no integer number of k splits fills 256 CUs x 2 blocks evenly (5 splits = 350 blocks left a third of the
// chip idle half of the time).  Pieces are stored in the accumulators' own lane order (1 KB coalesced per store) and
// added in k order by the reduce kernel: deterministic, no atomics.
template <int NT>
__global__ __launch_bounds__(kKtThreads) void gemm_ktile_mfma_kernel(
    const float* __restrict__ in, int64_t ld_in, const int32_t* __restrict__ idx, int32_t n_rows,
    const float* __restrict__ w, int32_t n_chunks, int32_t per, int32_t n_units, const float* __restrict__ bias,
    float* __restrict__ out, int64_t ld_out, float* __restrict__ pieces, int32_t s_max) {
  constexpr int NTP = NT == 3 ? 4 : NT;
  constexpr int N = 32 * NT;
  constexpr int kChunkFloats = 32 * 32 * NTP;
  constexpr int PAIRS = (32 * 32) / kKtThreads;          // (k, r) pairs of a chunk per thread
  __shared__ __attribute__((aligned(16))) float wl[2][kChunkFloats];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r_lo = lane & 31, khalf = lane >> 5;
  const int u0 = blockIdx.x * per, u1 = min(n_units, u0 + per);
  if (u0 >= u1) return;
  const int n_tiles = (n_rows + 31) >> 5;

  auto operand_rows = [&](int g) -> const float4* {      // this lane's operand row of row group g
    const int s_a = min((g * 8 + wave) * 32 + r_lo, n_rows - 1);
    const int64_t row = idx ? idx[s_a] : s_a;
    return reinterpret_cast<const float4*>(in + row * ld_in) + khalf * 4;
  };
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

  int g = u0 / n_chunks, c = u0 - g * n_chunks;
  const float4* src = operand_rows(g);
  float4 a_next[4];
  load_w(c);
#pragma unroll
  for (int i = 0; i < 4; ++i) a_next[i] = src[c * 8 + i];
  stash_w(0);
  __syncthreads();
  int cur = 0;
  for (int u = u0; u < u1; ++u) {
    float4 a4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a4[i] = a_next[i];
    const bool more = u + 1 < u1;
    const bool last_of_group = c + 1 == n_chunks;
    const int gn = last_of_group ? g + 1 : g, cn = last_of_group ? 0 : c + 1;
    const float4* srcn = src;
    if (more) {                                          // next unit's operands in flight behind this unit's MFMAs
      if (last_of_group) srcn = operand_rows(gn);
#pragma unroll
      for (int i = 0; i < 4; ++i) a_next[i] = srcn[cn * 8 + i];
      load_w(cn);
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
    if (last_of_group || !more) {
      // D[i][j]: j = lane & 31 = sample, feature 32 t + 8 q + 4 khalf + c in acc[t][4 q + c] (as rows_gemm.hip)
      if (pieces) {                                      // piece number = blocks since the one holding the group's first unit
        const int piece = blockIdx.x - (g * n_chunks) / per;
        float4* dst = reinterpret_cast<float4*>(pieces + ((int64_t)g * s_max + piece) * (256 * N)) + wave * (NT * 4 * 64) + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            dst[(t * 4 + q) * 64] = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
      } else {
        const int tile = g * 8 + wave, s_a = tile * 32 + r_lo;
        if (tile < n_tiles && s_a < n_rows) {
          const int64_t orow = idx ? idx[s_a] : s_a;
          float* dst = out + orow * ld_out + 4 * khalf;
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float4 v = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
              if (bias) v = f4_add(v, *reinterpret_cast<const float4*>(bias + 32 * t + 8 * q + 4 * khalf));
              *reinterpret_cast<float4*>(dst + 32 * t + 8 * q) = v;
            }
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }
    g = gn; c = cn; src = srcn;
  }
}

// out[row, :] = bias + the pieces of the row's group in k order.  One thread per float4 of the piece layout
// (coalesced reads of every piece; the 16-byte results go to their rows).
__global__ __launch_bounds__(256) void gemm_ktile_reduce_kernel(const float* __restrict__ pieces, int32_t s_max, int32_t n_chunks,
                                                                int32_t per, int32_t nt, const int32_t* __restrict__ idx,
                                                                int32_t n_rows, const float* __restrict__ bias,
                                                                float* __restrict__ out, int64_t ld_out) {
  const int per_group = 8 * nt * 4 * 64;                 // float4s of one piece
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int g = (int)(e / per_group), v = (int)(e % per_group);
  const int lane = v & 63, tq = (v >> 6) % (nt * 4), wave = v / (nt * 4 * 64);
  const int s_a = (g * 8 + wave) * 32 + (lane & 31);
  if (s_a >= n_rows) return;
  const int col = 32 * (tq >> 2) + 8 * (tq & 3) + 4 * (lane >> 5);
  const int p_first = (g * n_chunks) / per, p_last = ((g + 1) * n_chunks - 1) / per;
  const float4* p = reinterpret_cast<const float4*>(pieces + (int64_t)g * s_max * (256 * 32 * nt)) + v;
  float4 acc = bias ? *reinterpret_cast<const float4*>(bias + col) : f4_zero();
  const int n_p = p_last - p_first + 1;
  int k = 0;
  for (; k + 2 <= n_p; k += 2) {
    const float4 a = p[(int64_t)k * per_group], b = p[(int64_t)(k + 1) * per_group];
    acc = f4_add(f4_add(acc, a), b);
  }
  if (k < n_p) acc = f4_add(acc, p[(int64_t)k * per_group]);
  const int64_t row = idx ? idx[s_a] : s_a;
  *reinterpret_cast<float4*>(out + row * ld_out + col) = acc;
}

struct KtileGeometry { int groups, chunks, per, blocks, s_max; };   // s_max == 0: whole groups per block, no pieces

static KtileGeometry ktile_geometry(int32_t n_rows, int32_t k) {
  KtileGeometry q;
  q.groups = (n_rows + 255) / 256;
  q.chunks = k / 32;
  const int64_t units = (int64_t)q.groups * q.chunks;
  // 2 blocks per CU (16 waves) sustain the matrix cores best; when that leaves fewer than 12 chunks behind every
  // 128 KB piece (M = 17,716, K = 1,664: 8), one block per CU halves the piece traffic and wins (88 vs 93 us)
  static const int forced = [] { const char* e = getenv("GD_KTILE_SLOTS"); return e ? atoi(e) : 0; }();
  int per = (int)((units + 511) / 512);
  if (per < 12) per = (int)((units + 255) / 256);
  if (forced > 0) per = (int)((units + forced - 1) / forced);
  if (per < 4) per = 4;                                  // a piece costs a 128 KB round trip: at least 4 chunks of work behind it
  if (per >= q.chunks) {
    per = (per + q.chunks - 1) / q.chunks * q.chunks;
    q.s_max = 0;
  } else {
    q.s_max = (q.chunks + per - 2) / per + 1;
  }
  q.per = per;
  q.blocks = (int)((units + per - 1) / per);
  return q;
}

}  // namespace gd

extern "C" int64_t gd_gemm_f32_workspace(int32_t n_rows, int32_t k, int32_t n) {
  if (n_rows <= 0 || k <= 0 || k % 32) return 0;
  const gd::KtileGeometry q = gd::ktile_geometry(n_rows, k);
  return (int64_t)q.groups * q.s_max * 256 * n;
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
  const KtileGeometry q = ktile_geometry(n_rows, k);
  GD_REQUIRE(q.s_max == 0 || (workspace && aligned16(workspace)), GD_E_NULL, "gd_gemm_f32: workspace of gd_gemm_f32_workspace() floats");
  const dim3 grid(q.blocks), block(kKtThreads);
  float* pieces = q.s_max ? workspace : nullptr;
#define GD_KT_CASE(NT) \
  hipLaunchKernelGGL((gemm_ktile_mfma_kernel<NT>), grid, block, 0, s, in, ld_in, idx, n_rows, w, q.chunks, q.per, q.groups * q.chunks, \
                     bias, out, ld_out, pieces, q.s_max)
  switch (n / 32) {
    case 1: GD_KT_CASE(1); break;
    case 2: GD_KT_CASE(2); break;
    case 3: GD_KT_CASE(3); break;
    default: GD_KT_CASE(4); break;
  }
#undef GD_KT_CASE
  int rc = launched("gemm_ktile");
  if (rc || !q.s_max) return rc;
  const int64_t total = (int64_t)q.groups * 8 * (n / 32) * 4 * 64;
  hipLaunchKernelGGL(gemm_ktile_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, workspace, q.s_max, q.chunks, q.per,
                     n / 32, idx, n_rows, bias, out, ld_out);
  return launched("gemm_ktile_reduce");
}
