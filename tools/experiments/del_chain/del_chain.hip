// First-layer Del operator and the next layer's Linear in one pass over the rows of conv1's output:
//
//     z1[r,:] = pre[r,:] @ W_D                       r in S1            DeletionLayer.forward (deletion.py:17-29)
//     t2[r,:] = relu(z1[r,:] or pre[r,:]) @ W2^T      every row r        F.relu + the Linear of GCNConv / GINConv (gcn.py:17-21)
//
// Separately (gd_rows_gemm_signs_f32, then gd_rows_gemm_select_f32) the second product reads back the whole [N, d] matrix
// the first one has just written or passed over: 120 MB of the bench step's 372 MB for these two stages, and a launch.
// Here the second product takes the first one's accumulators as its operand, the way the fused last-layer kernel chains its
// products (del_fused.hip): after z^T = W_D^T x^T a lane holds, for its own sample, the features
// i = 32t + (r&3) + 8(r>>2) + 4kh in acc[t][r]; with exactly that feature as k slot (r, kh) of the second product the
// accumulator registers ARE its "B" operand and W2[c][i] comes out of a row-major LDS image as 16-byte reads.
// A row outside S1 skips the first product: its features are loaded straight into the accumulator layout (16-byte pieces).
//
// One 16-wave block per CU (the 64 KB interleaved W_D image of rows_gemm.hip + the 34 KB W2 image), tiles of 32 rows handed out
// through an LDS counter as in the row GEMMs; S1 tiles first (two products each), then the other rows' tiles.
#include "common.h"

namespace gd {

using f32x16c = __attribute__((ext_vector_type(16))) float;

template <int NO>      // NO = d_out2 / 32 (1 or 2); the Del width is 128 (NT = 4)
__global__ __launch_bounds__(1024, 4) void del_chain_kernel(const float* __restrict__ pre, int64_t ld_pre,
                                                            const int32_t* __restrict__ idx1, int32_t n1,
                                                            const int32_t* __restrict__ idx0, int32_t n0,
                                                            const float* __restrict__ w_del, const float* __restrict__ w2,
                                                            float* __restrict__ z1, int64_t ld_z1,
                                                            uint32_t* __restrict__ sign_out, float* __restrict__ t2,
                                                            int64_t ld_t2) {
  constexpr int NT = 4, D = 128, O = 32 * NO, PB = D + 4;
  extern __shared__ __attribute__((aligned(16))) float wl[];
  float* const w1i = wl;                  // w1i[(k * 32 + r) * 4 + t] = W_D[k][32 t + r]
  float* const w2i = wl + D * D;          // w2i[c * PB + k] = W2[c][k]
  __shared__ int q_next;
  const int tid = threadIdx.x, lane = tid & 63;
  const int r_lo = lane & 31, khalf = lane >> 5;
  for (int e = tid; e < D * 32; e += 1024) {
    const int k = e >> 5, r = e & 31;
    *reinterpret_cast<float4*>(w1i + e * 4) =
        make_float4(w_del[k * D + r], w_del[k * D + 32 + r], w_del[k * D + 64 + r], w_del[k * D + 96 + r]);
  }
  for (int e = tid; e < O * (D / 4); e += 1024) {
    const int c = e / (D / 4), k4 = e % (D / 4);
    *reinterpret_cast<float4*>(w2i + c * PB + 4 * k4) = reinterpret_cast<const float4*>(w2)[e];
  }
  if (tid == 0) q_next = 0;
  __syncthreads();

  const int nt1 = (n1 + 31) >> 5, nt0 = (n0 + 31) >> 5, n_tiles = nt1 + nt0;
  const int per_block = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_lo = blockIdx.x * per_block, t_hi = min(n_tiles, t_lo + per_block);
  auto grab = [&]() -> int {
    int t = 0;
    if (lane == 0) t = atomicAdd(&q_next, 1);
    t = __builtin_amdgcn_readfirstlane(t) + t_lo;
    return t < t_hi ? t : n_tiles;
  };
  const float* const w_bwd = w2i + r_lo * PB + 4 * khalf;

  // second product + store: t2^T tile = W2 relu(z)^T, k slot (r, kh) of k tile t <-> feature 32t + (r&3) + 8(r>>2) + 4kh
  auto second = [&](f32x16c (&acc)[NT], int32_t row, bool live) {
    f32x16c dacc[NO];
#pragma unroll
    for (int c = 0; c < NO; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) dacc[c][r] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = fmaxf(acc[t][r], 0.f);      // (one batch: not a VALU write in front of every matrix instruction)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int c = 0; c < NO; ++c) {
          const float4 f = *reinterpret_cast<const float4*>(w_bwd + 32 * c * PB + 32 * t + 8 * m);
          dacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, acc[t][4 * m + 0], dacc[c], 0, 0, 0);
          dacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, acc[t][4 * m + 1], dacc[c], 0, 0, 0);
          dacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, acc[t][4 * m + 2], dacc[c], 0, 0, 0);
          dacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, acc[t][4 * m + 3], dacc[c], 0, 0, 0);
        }
    if (live) {
      float* orow = t2 + (int64_t)row * ld_t2 + 4 * khalf;
#pragma unroll
      for (int c = 0; c < NO; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(orow + 32 * c + 8 * q) =
              make_float4(dacc[c][4 * q], dacc[c][4 * q + 1], dacc[c][4 * q + 2], dacc[c][4 * q + 3]);
    }
  };

  auto row_of = [&](int tile_) -> int32_t {
    if (tile_ >= n_tiles) return 0;
    if (tile_ < nt1) return idx1[min(tile_ * 32 + r_lo, n1 - 1)];
    return idx0[min((tile_ - nt1) * 32 + r_lo, n0 - 1)];
  };
  int tile = grab();
  if (tile >= n_tiles) return;
  int tile_nxt = grab();
  int32_t row = row_of(tile), row_nxt = row_of(tile_nxt);
  float4 a_next[4];
  if (tile < nt1) {
    const float4* src0 = reinterpret_cast<const float4*>(pre + (int64_t)row * ld_pre) + khalf * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) a_next[i] = src0[i];
  }
  for (; tile < n_tiles;) {
    // (the weight fragments are the same for every tile: keep them as loads inside the loop, see del_fused.hip)
    asm volatile("" ::: "memory");
    f32x16c acc[NT];
    const float4* src_n = reinterpret_cast<const float4*>(pre + (int64_t)row_nxt * ld_pre) + khalf * 4;
    if (tile < nt1) {
      // ---- S1 rows: z^T tile = W_D^T x^T, operand rows streamed in 32-wide k chunks (64 contiguous bytes per lane and
      // chunk, k slots {32 kc + 16 khalf + j}: rows_gemm.hip), the next chunk in flight while this one feeds the matrix pipe
      const int s_a = tile * 32 + r_lo;
      const bool live = s_a < n1;
      const float4* src = reinterpret_cast<const float4*>(pre + (int64_t)row * ld_pre) + khalf * 4;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
      for (int kc = 0; kc < D / 32; ++kc) {
        float4 a4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a4[i] = a_next[i];
        if (kc + 1 < D / 32) {
#pragma unroll
          for (int i = 0; i < 4; ++i) a_next[i] = src[(kc + 1) * 8 + i];
        } else if (tile_nxt < nt1) {          // (wave-uniform) the next S1 tile's first chunk: in flight during the second product
#pragma unroll
          for (int i = 0; i < 4; ++i) a_next[i] = src_n[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float av[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
          const int k0 = kc * 32 + khalf * 16 + i * 4;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float4 f = *reinterpret_cast<const float4*>(w1i + ((k0 + s) * 32 + r_lo) * 4);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, av[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, av[s], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, av[s], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, av[s], acc[3], 0, 0, 0);
          }
        }
      }
      // z1 row + the packed [z1 > 0] pattern (bit b of word t = feature 32t + b; the two half-row lanes merge their bits)
      float* dst = z1 + (int64_t)row * ld_z1 + 4 * khalf;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        uint32_t pos = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
          pos |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << (8 * q + 4 * khalf);
          if (live) *reinterpret_cast<float4*>(dst + 32 * t + 8 * q) = v;
        }
        pos |= (uint32_t)__shfl_xor((int)pos, 32);
        if (live && khalf == 0) sign_out[(int64_t)s_a * NT + t] = pos;
      }
      second(acc, row, live);
    } else {
      // ---- the other rows: their features straight into the accumulator layout
      const int s_a = (tile - nt1) * 32 + r_lo;
      const bool live = s_a < n0;
      const float* srow = pre + (int64_t)row * ld_pre + 4 * khalf;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(srow + 32 * t + 8 * q);
          acc[t][4 * q] = v.x; acc[t][4 * q + 1] = v.y; acc[t][4 * q + 2] = v.z; acc[t][4 * q + 3] = v.w;
        }
      second(acc, row, live);
    }
    row = row_nxt;
    tile = tile_nxt;
    tile_nxt = tile < n_tiles ? grab() : n_tiles;
    row_nxt = row_of(tile_nxt);
  }
}

}  // namespace gd

extern "C" int gd_del_chain_f32(const float* pre, int64_t ld_pre, const int32_t* idx1, int32_t n1, const int32_t* idx0,
                                int32_t n0, const float* w_del, const float* w2, float* z1, int64_t ld_z1,
                                uint32_t* sign_out, float* t2, int64_t ld_t2, int32_t d, int32_t d_out2, void* stream) {
  using namespace gd;
  if (n1 + n0 == 0) return GD_OK;
  GD_REQUIRE(pre && w_del && w2 && t2 && (n1 == 0 || (idx1 && z1 && sign_out)) && (n0 == 0 || idx0), GD_E_NULL,
             "gd_del_chain_f32: null pointer");
  GD_REQUIRE(n1 >= 0 && n0 >= 0 && d == 128 && (d_out2 == 32 || d_out2 == 64), GD_E_DIM,
             "gd_del_chain_f32: built for a 128-wide Del operator feeding a 32- / 64-wide Linear (d=%d, d_out2=%d)", d, d_out2);
  GD_REQUIRE(ld_pre >= d && ld_z1 >= d && ld_t2 >= d_out2 && ld_pre % 4 == 0 && ld_z1 % 4 == 0 && ld_t2 % 4 == 0, GD_E_DIM,
             "gd_del_chain_f32: bad row strides");
  GD_REQUIRE(aligned16(pre) && aligned16(z1) && aligned16(t2) && aligned16(w2), GD_E_ALIGN, "gd_del_chain_f32: unaligned");
  GD_REQUIRE(pre != z1 && pre != t2 && z1 != t2, GD_E_DIM, "gd_del_chain_f32: buffers must not alias");
  hipStream_t s = (hipStream_t)stream;
  const int n_tiles = (n1 + 31) / 32 + (n0 + 31) / 32;
  int grid = (n_tiles + 15) / 16;
  if (grid > 256) grid = 256;
#define GD_DC_CASE(NO_)                                                                                                   \
  do {                                                                                                                    \
    constexpr int kLds = (128 * 128 + 32 * NO_ * 132) * 4;                                                                \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&del_chain_kernel<NO_>),             \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds);                 \
    if (attr != hipSuccess) return fail(-(int)attr, "gd_del_chain_f32: %s", hipGetErrorString(attr));                     \
    hipLaunchKernelGGL((del_chain_kernel<NO_>), dim3(grid), dim3(1024), kLds, s, pre, ld_pre, idx1, n1, idx0, n0, w_del,  \
                       w2, z1, ld_z1, sign_out, t2, ld_t2);                                                               \
  } while (0)
  if (d_out2 == 32) GD_DC_CASE(1); else GD_DC_CASE(2);
#undef GD_DC_CASE
  return launched("del_chain");
}
