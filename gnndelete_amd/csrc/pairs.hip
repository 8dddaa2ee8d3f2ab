// Edge-probability Neighborhood-Influence term of GNNDeleteTrainer (framework/trainer/gnndelete.py:
// 174-193, 239-241) without the N x N logit matrix: over the |S| x |S| block of 2-hop S_Df nodes
//
//     L    = 1/count * sum_{valid (i > j)} ( sigmoid(z_i . z_j) - T[i][j] )^2
//     dz_i = 2/count * sum_{j : valid {i,j}} ( p_ij - t_ij ) p_ij (1 - p_ij) z_j
//
// value AND gradient in one pass over 32 x 32 tiles of pairs, two fp32 MFMA products per tile
// (flash-attention shaped): S^T = Z_J Z_I^T on the matrix cores, the elementwise sigmoid / residual /
// gate in the accumulator registers, then dZ_I^T += Z_J^T G^T on the matrix cores again with the
// gate tile fed straight from those registers (the MFMA k index of the second product is permuted to
// the accumulator layout of the first, so no transpose through LDS is needed).  Nothing of size
// |S|^2 is written; the only |S|^2 read is the target matrix, once per orientation.
//
// A block = 4 waves, each owning one 32-row tile I (its z rows live in registers); the block walks a
// range of column tiles J staged in LDS in two layouts ([k][j] for the first product, [j][c] for the
// second).  The J range is split over gridDim.y (split-K): partial dZ slabs are summed in a fixed
// order by a second kernel, so the result is deterministic.
#include "common.h"

namespace gd {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kPairsMaxSplit = 16;

// KD = d / 32
template <int KD>
__global__ __launch_bounds__(256, 2) void pairs_sigmoid_mse_kernel(
    const float* __restrict__ z, int64_t ld_z, const int32_t* __restrict__ nodes, int32_t n_s,
    const float* __restrict__ target, int64_t ld_t, float coef, int32_t tiles_per_split, int32_t n_pad,
    float* __restrict__ dz_part, float* __restrict__ loss_part) {
  constexpr int D = 32 * KD, HALF = D / 2;
  constexpr int PITCH = 33;                         // [k][j] image: +1 pad keeps the transposed fill cheap
  __shared__ float zj_kj[D * PITCH];
  __shared__ __attribute__((aligned(16))) float zj_jc[32 * D];
  __shared__ float red[4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, kh = lane >> 5;
  const int n_tiles = (n_s + 31) >> 5;
  const int it = blockIdx.x * 4 + wave;             // this wave's row tile (may be past the end: masked)
  const int gi = it * 32 + lo;                      // this lane's pair row i (index into S)
  const bool i_ok = gi < n_s;

  // ---- this lane's half of its z row: MFMA k slot (kk, kh) <-> feature kh * HALF + kk
  float zi[HALF];
  {
    const int64_t row = nodes[min(gi, n_s - 1)];
    const float4* src = reinterpret_cast<const float4*>(z + row * ld_z + kh * HALF);
#pragma unroll
    for (int q = 0; q < HALF / 4; ++q) {
      const float4 v = src[q];
      zi[4 * q] = v.x; zi[4 * q + 1] = v.y; zi[4 * q + 2] = v.z; zi[4 * q + 3] = v.w;
    }
  }

  f32x16 dacc[KD];
#pragma unroll
  for (int c = 0; c < KD; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[c][r] = 0.f;
  float loss = 0.f;

  const int jt0 = blockIdx.y * tiles_per_split;
  const int jt1 = min(n_tiles, jt0 + tiles_per_split);
  for (int jt = jt0; jt < jt1; ++jt) {
    // ---- stage Z_J (32 rows x D) in both layouts
    __syncthreads();
    for (int f = tid; f < 32 * (D / 4); f += 256) {
      const int j = f / (D / 4), c4 = f % (D / 4);
      const int64_t row = nodes[min(jt * 32 + j, n_s - 1)];
      const float4 v = reinterpret_cast<const float4*>(z + row * ld_z)[c4];
      reinterpret_cast<float4*>(zj_jc)[j * (D / 4) + c4] = v;
      zj_kj[(4 * c4 + 0) * PITCH + j] = v.x;
      zj_kj[(4 * c4 + 1) * PITCH + j] = v.y;
      zj_kj[(4 * c4 + 2) * PITCH + j] = v.z;
      zj_kj[(4 * c4 + 3) * PITCH + j] = v.w;
    }
    __syncthreads();
    if (it >= n_tiles) continue;                    // (uniform per wave; barriers above stay matched)

    // ---- targets of this lane's 16 pairs, in flight during the first product.
    // accumulator element r of this lane is the pair (i = gi, j = jt*32 + jl(r)), jl(r) = (r&3) + 8(r>>2) + 4kh
    float tv[16];
    const int gj0 = jt * 32;
    if (jt < it) {                                  // strictly below the diagonal: T[i][j], 4 x float4 of own row
      const float* trow = target + (int64_t)min(gi, n_s - 1) * ld_t + gj0 + 4 * kh;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(trow + 8 * q);
        tv[4 * q] = v.x; tv[4 * q + 1] = v.y; tv[4 * q + 2] = v.z; tv[4 * q + 3] = v.w;
      }
    } else {                                        // above / on the diagonal: T[max][min], lanes coalesce over i
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gj = min(gj0 + (r & 3) + 8 * (r >> 2) + 4 * kh, n_s - 1);
        const int a = max(min(gi, n_s - 1), gj), b = min(min(gi, n_s - 1), gj);
        tv[r] = target[(int64_t)a * ld_t + b];
      }
    }

    // ---- S^T tile = Z_J Z_I^T
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < HALF; ++kk)
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(zj_kj[(kh * HALF + kk) * PITCH + lo], zi[kk], sacc, 0, 0, 0);

    // ---- residual, loss, gate (in place in the accumulator registers)
    float gv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gj = gj0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      const float t = tv[r];
      const bool ok = i_ok && gj < n_s && gj != gi && t >= 0.f;
      const float p = 1.0f / (1.0f + expf(-sacc[r]));
      const float df = p - t;
      gv[r] = ok ? coef * df * p * (1.0f - p) : 0.f;
      if (ok && gi > gj) loss = fmaf(df, df, loss);
    }

    // ---- dZ_I^T += Z_J^T G^T : k slot (kk, kh) <-> j = (kk&3) + 8(kk>>2) + 4kh, i.e. gate element kk
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int jl = (kk & 3) + 8 * (kk >> 2) + 4 * kh;
#pragma unroll
      for (int c = 0; c < KD; ++c)
        dacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(zj_jc[jl * D + 32 * c + lo], gv[kk], dacc[c], 0, 0, 0);
    }
  }

  // ---- partial dZ slab: lane holds column i = lo and features 32c + 8q + 4kh + (0..3)
  if (it < n_tiles) {
    float* dst = dz_part + ((int64_t)blockIdx.y * n_pad + it * 32 + lo) * D + 4 * kh;
#pragma unroll
    for (int c = 0; c < KD; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(dst + 32 * c + 8 * q) =
            make_float4(dacc[c][4 * q], dacc[c][4 * q + 1], dacc[c][4 * q + 2], dacc[c][4 * q + 3]);
  }
  loss = wave_sum(loss);
  if (lane == 0) red[wave] = loss;
  __syncthreads();
  if (tid == 0) loss_part[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// dz[i,:] = sum over splits (fixed order); block 0 also folds the loss partials: loss = inv_count * sum
__global__ __launch_bounds__(256) void pairs_reduce_kernel(const float* __restrict__ dz_part, int32_t n_split,
                                                           int32_t n_pad, int32_t n_s, int32_t d4,
                                                           float* __restrict__ dz, const float* __restrict__ loss_part,
                                                           int32_t n_loss_part, float inv_count,
                                                           float* __restrict__ loss_out) {
  __shared__ float red[256];
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < (int64_t)n_s * d4) {
    float4 s = f4_zero();
    for (int k = 0; k < n_split; ++k)
      s = f4_add(s, reinterpret_cast<const float4*>(dz_part + (int64_t)k * n_pad * d4 * 4)[e]);
    reinterpret_cast<float4*>(dz)[e] = s;
  }
  if (blockIdx.x == 0) {
    float a = 0.f;
    for (int i = threadIdx.x; i < n_loss_part; i += 256) a += loss_part[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) *loss_out = red[0] * inv_count;
  }
}

// generic feature width: one thread per pair row i, plain loops (small problems / odd dims only)
__global__ __launch_bounds__(64) void pairs_scalar_kernel(const float* __restrict__ z, int64_t ld_z,
                                                          const int32_t* __restrict__ nodes, int32_t n_s, int32_t d,
                                                          const float* __restrict__ target, int64_t ld_t, float coef,
                                                          float* __restrict__ dz, float* __restrict__ loss_part) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  float loss = 0.f;
  if (i < n_s) {
    const float* zi = z + (int64_t)nodes[i] * ld_z;
    float* di = dz + (int64_t)i * d;
    for (int c = 0; c < d; ++c) di[c] = 0.f;
    for (int j = 0; j < n_s; ++j) {
      if (j == i) continue;
      const float t = target[(int64_t)max(i, j) * ld_t + min(i, j)];
      if (!(t >= 0.f)) continue;
      const float* zj = z + (int64_t)nodes[j] * ld_z;
      float s = 0.f;
      for (int c = 0; c < d; ++c) s = fmaf(zi[c], zj[c], s);
      const float p = 1.0f / (1.0f + expf(-s));
      const float df = p - t;
      if (i > j) loss = fmaf(df, df, loss);
      const float g = coef * df * p * (1.0f - p);
      for (int c = 0; c < d; ++c) di[c] = fmaf(g, zj[c], di[c]);
    }
  }
  loss = wave_sum(loss);
  if (threadIdx.x == 0) loss_part[blockIdx.x] = loss;
}

static inline void pairs_geometry(int32_t n_s, int* row_blocks, int* n_split, int* tiles_per_split, int* n_pad) {
  const int n_tiles = (n_s + 31) / 32;
  *row_blocks = (n_tiles + 3) / 4;
  int js = (1024 + *row_blocks - 1) / (*row_blocks > 0 ? *row_blocks : 1);
  if (js > kPairsMaxSplit) js = kPairsMaxSplit;
  if (js > n_tiles) js = n_tiles;
  if (js < 1) js = 1;
  *tiles_per_split = (n_tiles + js - 1) / js;
  *n_split = (n_tiles + *tiles_per_split - 1) / (*tiles_per_split > 0 ? *tiles_per_split : 1);
  if (*n_split < 1) *n_split = 1;
  *n_pad = n_tiles * 32;
}

static inline bool pairs_mfma_ok(int32_t d) { return d == 32 || d == 64 || d == 128; }

}  // namespace gd

extern "C" int64_t gd_pairs_sigmoid_mse_workspace(int32_t n_s, int32_t d) {
  using namespace gd;
  if (n_s <= 0 || d <= 0) return 1;
  if (!pairs_mfma_ok(d)) return (n_s + 63) / 64 + 1;
  int rb, js, tps, n_pad;
  pairs_geometry(n_s, &rb, &js, &tps, &n_pad);
  return (int64_t)js * n_pad * d + (int64_t)js * rb + 4;
}

extern "C" int gd_pairs_sigmoid_mse_f32(const float* z, int64_t ld_z, const int32_t* nodes, int32_t n_s, int32_t d,
                                        const float* target, int64_t ld_t, float inv_count, float* loss, float* dz,
                                        float* workspace, void* stream) {
  using namespace gd;
  GD_REQUIRE(loss && workspace, GD_E_NULL, "gd_pairs_sigmoid_mse_f32: null output");
  GD_REQUIRE(n_s >= 0 && d > 0 && ld_z >= d && ld_t >= n_s, GD_E_DIM, "gd_pairs_sigmoid_mse_f32: bad dims");
  GD_REQUIRE(n_s == 0 || (z && nodes && target && dz), GD_E_NULL, "gd_pairs_sigmoid_mse_f32: null input");
  hipStream_t s = (hipStream_t)stream;
  const float coef = 2.0f * inv_count;
  if (n_s == 0) {
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), s);
    if (e != hipSuccess) return fail(-(int)e, "gd_pairs_sigmoid_mse_f32: %s", hipGetErrorString(e));
    return GD_OK;
  }
  const bool vec = pairs_mfma_ok(d) && aligned16(z) && ld_z % 4 == 0 && aligned16(target) && ld_t % 4 == 0 &&
                   aligned16(dz) && aligned16(workspace);
  if (!vec) {
    const int nb = (n_s + 63) / 64;
    hipLaunchKernelGGL(pairs_scalar_kernel, dim3(nb), dim3(64), 0, s, z, ld_z, nodes, n_s, d, target, ld_t, coef, dz,
                       workspace);
    int rc = launched("pairs_scalar");
    if (rc) return rc;
    hipLaunchKernelGGL(pairs_reduce_kernel, dim3(1), dim3(256), 0, s, nullptr, 0, 0, 0, 0, dz, workspace, nb,
                       inv_count, loss);
    return launched("pairs_reduce");
  }
  int rb, js, tps, n_pad;
  pairs_geometry(n_s, &rb, &js, &tps, &n_pad);
  float* dz_part = workspace;
  float* loss_part = workspace + (int64_t)js * n_pad * d;
  const dim3 grid(rb, js);
  switch (d / 32) {
    case 1:
      hipLaunchKernelGGL((pairs_sigmoid_mse_kernel<1>), grid, dim3(256), 0, s, z, ld_z, nodes, n_s, target, ld_t, coef,
                         tps, n_pad, dz_part, loss_part);
      break;
    case 2:
      hipLaunchKernelGGL((pairs_sigmoid_mse_kernel<2>), grid, dim3(256), 0, s, z, ld_z, nodes, n_s, target, ld_t, coef,
                         tps, n_pad, dz_part, loss_part);
      break;
    default:
      hipLaunchKernelGGL((pairs_sigmoid_mse_kernel<4>), grid, dim3(256), 0, s, z, ld_z, nodes, n_s, target, ld_t, coef,
                         tps, n_pad, dz_part, loss_part);
      break;
  }
  int rc = launched("pairs_sigmoid_mse");
  if (rc) return rc;
  const int64_t n4 = (int64_t)n_s * (d / 4);
  hipLaunchKernelGGL(pairs_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, dz_part, js, n_pad, n_s,
                     d / 4, dz, loss_part, js * rb, inv_count, loss);
  return launched("pairs_reduce");
}
