// Fused Deleted-Edge-Consistency + Neighborhood-Influence MSE terms (value AND gradient in one
// pass), link decoders, Adam.  All HBM/latency-bound row-vector kernels: LPR lanes x float4
// cover a row, 64/LPR rows (segments / edges) per wave, wave-level xor-shuffle reductions, one
// partial per block, partials combined by a single block in a fixed order (deterministic).
#include "common.h"

namespace gd {

template <int LPR, int VPL>
__global__ __launch_bounds__(256) void rowpair_mse_kernel(
    const float* __restrict__ z, int64_t ld_z, const float* __restrict__ o, int64_t ld_o, int32_t d4,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ seg_row, int32_t n_seg,
    const int32_t* __restrict__ term_o, const float* __restrict__ term_w, const int32_t* __restrict__ term_kind,
    float* __restrict__ dz, int64_t ld_dz, int32_t dz_compact, float* __restrict__ partials) {
  constexpr int G = kWave / LPR;
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane / LPR, li = lane % LPR;
  const int u = (blockIdx.x * 4 + wave) * G + g;
  float s0 = 0.f, s1 = 0.f;
  if (u < n_seg) {
    const int row = seg_row[u];
    const int t0 = seg_ptr[u], t1 = seg_ptr[u + 1];
    float4 zr[VPL], gr[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int vec = li + v * LPR;
      zr[v] = vec < d4 ? reinterpret_cast<const float4*>(z + (int64_t)row * ld_z)[vec] : f4_zero();
      gr[v] = f4_zero();
    }
    for (int t = t0; t < t1; ++t) {
      const float4* orow = reinterpret_cast<const float4*>(o + (int64_t)term_o[t] * ld_o);
      const float w2 = 2.0f * term_w[t];
      float sq = 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int vec = li + v * LPR;
        if (vec >= d4) continue;
        const float4 ov = orow[vec];
        const float4 df = make_float4(zr[v].x - ov.x, zr[v].y - ov.y, zr[v].z - ov.z, zr[v].w - ov.w);
        sq = fmaf(df.x, df.x, sq); sq = fmaf(df.y, df.y, sq); sq = fmaf(df.z, df.z, sq); sq = fmaf(df.w, df.w, sq);
        gr[v] = f4_fma(w2, df, gr[v]);
      }
      if (term_kind[t] == 0) s0 += sq; else s1 += sq;
    }
    float* drow = dz + (int64_t)(dz_compact ? u : row) * ld_dz;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int vec = li + v * LPR;
      if (vec < d4) reinterpret_cast<float4*>(drow)[vec] = gr[v];
    }
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partials[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

__global__ __launch_bounds__(256) void pair_sum_reduce_kernel(const float* __restrict__ partials, int32_t n_part,
                                                              float* __restrict__ sums) {
  __shared__ float red[2][256];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < n_part; i += 256) { a += partials[2 * i]; b += partials[2 * i + 1]; }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { red[0][threadIdx.x] += red[0][threadIdx.x + off]; red[1][threadIdx.x] += red[1][threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { sums[0] += red[0][0]; sums[1] += red[1][0]; }
}

// ---------------------------------------------------------------------------------------------
// Regular ("row-target") form used when the targets are fixed for the whole run (full-batch
// training: z_ori and the negatives are drawn once, gnndelete_nodeemb.py:177-185).  All terms
// that touch row i are folded, once, into their mean target tbar_i, their count c_i and a
// constant:   sum_t |z_i - o_t|^2 = c_i |z_i - tbar_i|^2 + K_i     (exact, no cancellation)
// so the per-step work is a pure stream:  read z[row_u], read tm[u] (compact, sequential),
// write dz[row_u] = coef_u (z - tbar),  sums[kind_u] += cnt_u |z - tbar|^2.
// A wave takes 64 consecutive loss rows: one coalesced fetch of their (row, coef, cnt, kind),
// then its G lane groups walk them 4 rows at a time (8 row loads in flight per group).
struct RowTargetJob {
  const float* z; int64_t ld_z; const float* tm; int32_t d4;
  const int32_t* row_idx; const float* coef; const float* cnt; const int32_t* kind; int32_t n_rows;
  float* dz; int64_t ld_dz; float* partials;
};

template <int LPR, int VPL>
__device__ __forceinline__ void rowtarget_mse_body(
    const float* __restrict__ z, int64_t ld_z, const float* __restrict__ tm, int32_t d4,
    const int32_t* __restrict__ row_idx, const float* __restrict__ coef, const float* __restrict__ cnt,
    const int32_t* __restrict__ kind, int32_t n_rows, float* __restrict__ dz, int64_t ld_dz,
    float* __restrict__ partials, const int block) {
  constexpr int G = kWave / LPR;
  constexpr int U = (kWave / G) >= 4 ? 4 : (kWave / G);
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane / LPR, li = lane % LPR;
  const int u0 = (block * 4 + wave) * kWave;
  float s0 = 0.f, s1 = 0.f;
  if (u0 < n_rows) {
    const int n_here = min(kWave, n_rows - u0);
    const bool live = lane < n_here;
    const int row_l = live ? row_idx[u0 + lane] : 0;
    const float coef_l = live ? coef[u0 + lane] : 0.f;
    // count with the kind folded into the sign: DEC (kind 0) >= 0, NI (kind 1) stored negative
    const float cnt_l = live ? (kind[u0 + lane] ? -cnt[u0 + lane] : cnt[u0 + lane]) : 0.f;
    const int trips = (n_here + G - 1) / G;
    for (int t0 = 0; t0 < trips; t0 += U) {
      float4 zv[U][VPL], tv[U][VPL];
      int rows[U];
      float cf[U], cn[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = (t0 + u) * G + g;
        ok[u] = j < n_here;
        rows[u] = __shfl(row_l, j & 63);
        cf[u] = __shfl(coef_l, j & 63);
        cn[u] = __shfl(cnt_l, j & 63);
        const float4* zr = reinterpret_cast<const float4*>(z + (int64_t)rows[u] * ld_z);
        const float4* tr = reinterpret_cast<const float4*>(tm + (int64_t)(u0 + j) * d4 * 4);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int vec = li + v * LPR;
          const bool in = ok[u] && vec < d4;
          zv[u][v] = in ? zr[vec] : f4_zero();
          tv[u][v] = in ? tr[vec] : f4_zero();
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float sq = 0.f;
        float* drow = dz + (int64_t)rows[u] * ld_dz;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int vec = li + v * LPR;
          const float4 df = make_float4(zv[u][v].x - tv[u][v].x, zv[u][v].y - tv[u][v].y,
                                        zv[u][v].z - tv[u][v].z, zv[u][v].w - tv[u][v].w);
          sq = fmaf(df.x, df.x, sq); sq = fmaf(df.y, df.y, sq); sq = fmaf(df.z, df.z, sq); sq = fmaf(df.w, df.w, sq);
          if (dz && ok[u] && vec < d4)                 // dz == NULL: the loss value only (rows whose gradient feeds nothing)
            reinterpret_cast<float4*>(drow)[vec] = make_float4(cf[u] * df.x, cf[u] * df.y, cf[u] * df.z, cf[u] * df.w);
        }
        if (cn[u] >= 0.f) s0 = fmaf(cn[u], sq, s0); else s1 = fmaf(-cn[u], sq, s1);
      }
    }
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * block + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partials[2 * block + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

template <int LPR, int VPL>
__global__ __launch_bounds__(256) void rowtarget_mse_kernel(
    const float* __restrict__ z, int64_t ld_z, const float* __restrict__ tm, int32_t d4,
    const int32_t* __restrict__ row_idx, const float* __restrict__ coef, const float* __restrict__ cnt,
    const int32_t* __restrict__ kind, int32_t n_rows, float* __restrict__ dz, int64_t ld_dz,
    float* __restrict__ partials) {
  rowtarget_mse_body<LPR, VPL>(z, ld_z, tm, d4, row_idx, coef, cnt, kind, n_rows, dz, ld_dz, partials, blockIdx.x);
}

// Two row-target jobs of different widths in ONE launch (blocks [0, nb_a) walk job a, the rest job b): the two stand-alone
// loss launches of a knowledge-graph step - the layer-1 DEC rows (128 floats, loss sums only) and the layer-2 DEC rows
// (64 floats, gradient rows written) - are launch-sized on their own (293 blocks each).
template <int LA, int LB>
__global__ __launch_bounds__(256) void rowtarget_mse_pair_kernel(RowTargetJob a, RowTargetJob b, int32_t nb_a) {
  if ((int)blockIdx.x < nb_a)
    rowtarget_mse_body<LA, 1>(a.z, a.ld_z, a.tm, a.d4, a.row_idx, a.coef, a.cnt, a.kind, a.n_rows, a.dz, a.ld_dz, a.partials, blockIdx.x);
  else
    rowtarget_mse_body<LB, 1>(b.z, b.ld_z, b.tm, b.d4, b.row_idx, b.coef, b.cnt, b.kind, b.n_rows, b.dz, b.ld_dz, b.partials, blockIdx.x - nb_a);
}

static inline int mse_blocks(int32_t n_seg, int lpr) {
  const int per_block = 4 * (kWave / lpr);
  return (n_seg + per_block - 1) / per_block;
}

template <int LPR, int VPL, bool DISTMULT>
__global__ __launch_bounds__(256) void edge_dot_kernel(const float* __restrict__ z, int64_t ld_z, int32_t d4,
                                                       const int64_t* __restrict__ e0, const int64_t* __restrict__ e1,
                                                       const float* __restrict__ rel, int64_t ld_rel,
                                                       const int64_t* __restrict__ etype, int64_t n_edges,
                                                       float* __restrict__ out) {
  constexpr int G = kWave / LPR;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR, li = lane % LPR;
  const int64_t m = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * G + g;
  float p = 0.f;
  if (m < n_edges) {
    const float4* a = reinterpret_cast<const float4*>(z + e0[m] * ld_z);
    const float4* b = reinterpret_cast<const float4*>(z + e1[m] * ld_z);
    const float4* r = DISTMULT ? reinterpret_cast<const float4*>(rel + etype[m] * ld_rel) : nullptr;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int vec = li + v * LPR;
      if (vec >= d4) continue;
      float4 av = a[vec];
      const float4 bv = b[vec];
      if (DISTMULT) { const float4 rv = r[vec]; av.x *= rv.x; av.y *= rv.y; av.z *= rv.z; av.w *= rv.w; }
      p = fmaf(av.x, bv.x, p); p = fmaf(av.y, bv.y, p); p = fmaf(av.z, bv.z, p); p = fmaf(av.w, bv.w, p);
    }
  }
#pragma unroll
  for (int off = 1; off < LPR; off <<= 1) p += __shfl_xor(p, off);
  if (li == 0 && m < n_edges) out[m] = p;
}

// Decoder backward (input gradient): dz[v,:] = sum over the decoded edges incident to v of
//     w[k] * z[other[k],:]  (* rel[et[k],:] for DistMult)
// over a node-major incidence list inc_ptr[n+1] whose entries carry the OTHER endpoint, the upstream gradient of
// the edge and (DistMult) its relation - so an incidence costs one dependent gather, the z row.  One lane group per
// node, four incidences in flight, added in list order; a node with more than kHeavy incidences (a hub of a heavy-tailed
// graph: 2,000 positive edges at collab size, which alone took 1 ms in list order) is redone by the whole wave, its lane
// groups taking every G-th incidence and adding up in a fixed shuffle pattern.  No atomics, the same bits every run
// (autograd's two index_add_ calls add with atomics in arrival order).  Nodes without decoded edges get zeros.
template <int LPR, bool DISTMULT>
__global__ __launch_bounds__(256) void edge_dot_bwd_kernel(const float* __restrict__ z, int64_t ld_z, int32_t d4,
                                                           const int32_t* __restrict__ other, const float* __restrict__ wv,
                                                           const float* __restrict__ rel, int64_t ld_rel,
                                                           const int32_t* __restrict__ et,
                                                           const int64_t* __restrict__ inc_ptr, int64_t n_nodes,
                                                           float* __restrict__ dz, int64_t ld_dz) {
  constexpr int G = kWave / LPR;
  constexpr int kHeavy = 64;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR, li = lane % LPR;
  const int64_t base = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * G;
  const int64_t v = base + g;
  int64_t k0 = 0, k1 = 0;
  if (v < n_nodes) { k0 = inc_ptr[v]; k1 = inc_ptr[v + 1]; }
  // sum of the incidences first, first + step, ... < last of one node, this lane's float4 `vec`
  auto walk = [&](int64_t first, int64_t last, int step, int vec) -> float4 {
    auto term = [&](int64_t k) -> float4 {
      float4 a = reinterpret_cast<const float4*>(z + (int64_t)other[k] * ld_z)[vec];
      if (DISTMULT) {
        const float4 r = reinterpret_cast<const float4*>(rel + (int64_t)et[k] * ld_rel)[vec];
        a.x *= r.x; a.y *= r.y; a.z *= r.z; a.w *= r.w;
      }
      return a;
    };
    float4 acc = f4_zero();
    int64_t k = first;
    for (; k + 3 * step < last; k += 4 * step) {
      const float4 a0 = term(k), a1 = term(k + step), a2 = term(k + 2 * step), a3 = term(k + 3 * step);
      acc = f4_fma(wv[k], a0, acc);
      acc = f4_fma(wv[k + step], a1, acc);
      acc = f4_fma(wv[k + 2 * step], a2, acc);
      acc = f4_fma(wv[k + 3 * step], a3, acc);
    }
    for (; k < last; k += step) acc = f4_fma(wv[k], term(k), acc);
    return acc;
  };
  const bool heavy = G > 1 && k1 - k0 > kHeavy;
  if (v < n_nodes && !heavy)
    for (int vec = li; vec < d4; vec += LPR) reinterpret_cast<float4*>(dz + v * ld_dz)[vec] = walk(k0, k1, 1, vec);
  unsigned long long todo = __ballot(heavy && li == 0);
  while (todo) {                                           // wave-uniform: every lane group helps with each hub of the wave
    const int g2 = (__ffsll((long long)todo) - 1) / LPR;
    todo &= todo - 1;
    const int64_t v2 = base + g2;
    const int64_t h0 = inc_ptr[v2], h1 = inc_ptr[v2 + 1];
    for (int vec = li; vec < d4; vec += LPR) {
      float4 acc = walk(h0 + g, h1, G, vec);
#pragma unroll
      for (int off = LPR; off < kWave; off <<= 1) acc = f4_add(acc, f4_shfl_xor(acc, off));
      if (g == 0) reinterpret_cast<float4*>(dz + v2 * ld_dz)[vec] = acc;
    }
  }
}

template <bool DISTMULT>
__global__ __launch_bounds__(256) void edge_dot_scalar_kernel(const float* __restrict__ z, int64_t ld_z, int32_t d,
                                                              const int64_t* __restrict__ e0,
                                                              const int64_t* __restrict__ e1,
                                                              const float* __restrict__ rel, int64_t ld_rel,
                                                              const int64_t* __restrict__ etype, int64_t n_edges,
                                                              float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= n_edges) return;
  float p = 0.f;
  for (int c = lane; c < d; c += kWave) {
    float av = z[e0[m] * ld_z + c];
    if (DISTMULT) av *= rel[etype[m] * ld_rel + c];
    p = fmaf(av, z[e1[m] * ld_z + c], p);
  }
  p = wave_sum(p);
  if (lane == 0) out[m] = p;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   int32_t* __restrict__ step, int64_t n, double lr, double beta1,
                                                   double beta2, double eps) {
  // every thread reads the pre-increment counter; thread 0 of the LAST block bumps it.  The
  // grid is small (Del weights: 20k elements) so all blocks read before that block retires in
  // practice, but correctness must not rely on it: the bump is done by a separate kernel.
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const AdamScalars sc = adam_scalars(lr, beta1, beta2, eps, *step + 1);
  float pi = param[i], mi = m[i], vi = v[i];
  adam_update(pi, mi, vi, grad[i], sc);
  param[i] = pi; m[i] = mi; v[i] = vi;
}

__global__ void bump_step_kernel(int32_t* step) { *step += 1; }

// Adam whose step number is read from a shared iteration counter (t = *iter + 1, not modified)
__global__ __launch_bounds__(256) void adam_at_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      const int32_t* __restrict__ iter, int64_t n, double lr,
                                                      double beta1, double beta2, double eps) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const AdamScalars sc = adam_scalars(lr, beta1, beta2, eps, *iter + 1);
  float pi = param[i], mi = m[i], vi = v[i];
  adam_update(pi, mi, vi, grad[i], sc);
  param[i] = pi; m[i] = mi; v[i] = vi;
}

// One block: reduce the per-block partial sums of both layers' loss kernels in a fixed order, append
// the four sums (r1, l1, r2, l2) to the device-side history ring and advance the ring position and
// the iteration counter.  Replaces memset + 2 reductions + 3 tiny index kernels per step.
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ p1, int32_t n1,
                                                            const float* __restrict__ p2, int32_t n2,
                                                            const float* __restrict__ extra, float* __restrict__ hist,
                                                            int32_t capacity, int32_t* __restrict__ pos,
                                                            int32_t* __restrict__ iter) {
  __shared__ float red[4][256];
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  for (int i = threadIdx.x; i < n1; i += 256) { a += p1[2 * i]; b += p1[2 * i + 1]; }
  for (int i = threadIdx.x; i < n2; i += 256) { c += p2[2 * i]; d += p2[2 * i + 1]; }
  red[0][threadIdx.x] = a; red[1][threadIdx.x] = b; red[2][threadIdx.x] = c; red[3][threadIdx.x] = d;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
#pragma unroll
      for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) hist[(int64_t)(*pos) * 4 + threadIdx.x] = red[threadIdx.x][0] + (extra ? extra[threadIdx.x] : 0.f);
  __syncthreads();
  if (threadIdx.x == 0) {
    *pos = (*pos + 1) % capacity;
    *iter += 1;
  }
}

}  // namespace gd

extern "C" int64_t gd_rowpair_mse_workspace(int32_t n_seg) {
  // worst case LPR = 64 (one segment per wave): 4 segments per block, 2 floats per block
  return 2 * (int64_t)((n_seg + 3) / 4 + 1);
}

extern "C" int gd_rowpair_mse_f32(const float* z, int64_t ld_z, const float* o, int64_t ld_o, int32_t d,
                                  const int32_t* seg_ptr, const int32_t* seg_row, int32_t n_seg, const int32_t* term_o,
                                  const float* term_w, const int32_t* term_kind, float* dz, int64_t ld_dz,
                                  int32_t dz_compact, float* sums, float* partials, void* stream) {
  using namespace gd;
  GD_REQUIRE(sums && partials, GD_E_NULL, "gd_rowpair_mse_f32: null sums/partials");
  if (n_seg == 0) return GD_OK;
  GD_REQUIRE(z && o && seg_ptr && seg_row && term_o && term_w && term_kind && dz, GD_E_NULL,
             "gd_rowpair_mse_f32: null pointer");
  GD_REQUIRE(d > 0 && d % 4 == 0 && d <= 1024 && ld_z % 4 == 0 && ld_o % 4 == 0 && ld_dz % 4 == 0, GD_E_DIM,
             "gd_rowpair_mse_f32: d=%d must be a multiple of 4 (<=1024) with 16-byte row strides", d);
  GD_REQUIRE(aligned16(z) && aligned16(o) && aligned16(dz), GD_E_ALIGN, "gd_rowpair_mse_f32: unaligned matrix");
  hipStream_t s = (hipStream_t)stream;
  const int d4 = d / 4;
  const int lpr = lanes_per_row(d4);
  const int nb = mse_blocks(n_seg, lpr);
#define GD_MSE_CASE(LPR, VPL)                                                                                       \
  hipLaunchKernelGGL((rowpair_mse_kernel<LPR, VPL>), dim3(nb), dim3(256), 0, s, z, ld_z, o, ld_o, d4, seg_ptr,      \
                     seg_row, n_seg, term_o, term_w, term_kind, dz, ld_dz, dz_compact, partials)
  switch (lpr) {
    case 1: GD_MSE_CASE(1, 1); break;
    case 2: GD_MSE_CASE(2, 1); break;
    case 4: GD_MSE_CASE(4, 1); break;
    case 8: GD_MSE_CASE(8, 1); break;
    case 16: GD_MSE_CASE(16, 1); break;
    case 32: GD_MSE_CASE(32, 1); break;
    default:
      if (d4 <= 64) GD_MSE_CASE(64, 1);
      else if (d4 <= 128) GD_MSE_CASE(64, 2);
      else GD_MSE_CASE(64, 4);
  }
#undef GD_MSE_CASE
  int rc = launched("rowpair_mse");
  if (rc) return rc;
  hipLaunchKernelGGL(pair_sum_reduce_kernel, dim3(1), dim3(256), 0, s, partials, nb, sums);
  return launched("pair_sum_reduce");
}

extern "C" int64_t gd_rowtarget_mse_workspace(int32_t n_rows) { return 2 * (int64_t)((n_rows + 255) / 256 + 1); }

extern "C" int gd_rowtarget_mse_f32(const float* z, int64_t ld_z, const float* tm, int32_t d, const int32_t* row_idx,
                                    const float* coef, const float* cnt, const int32_t* kind, int32_t n_rows,
                                    float* dz, int64_t ld_dz, float* sums, float* partials, void* stream) {
  using namespace gd;
  GD_REQUIRE(partials, GD_E_NULL, "gd_rowtarget_mse_f32: null partials");
  if (n_rows == 0) return GD_OK;
  GD_REQUIRE(z && tm && row_idx && coef && cnt && kind, GD_E_NULL, "gd_rowtarget_mse_f32: null pointer");
  GD_REQUIRE(d > 0 && d % 4 == 0 && d <= 1024 && ld_z % 4 == 0 && ld_dz % 4 == 0, GD_E_DIM,
             "gd_rowtarget_mse_f32: d=%d must be a multiple of 4 (<=1024) with 16-byte row strides", d);
  GD_REQUIRE(aligned16(z) && aligned16(tm) && aligned16(dz), GD_E_ALIGN, "gd_rowtarget_mse_f32: unaligned matrix");      // (dz may be NULL)
  hipStream_t s = (hipStream_t)stream;
  const int d4 = d / 4;
  const int lpr = lanes_per_row(d4);
  const int nb = (n_rows + 255) / 256;
#define GD_RT_CASE(LPR, VPL)                                                                                      \
  hipLaunchKernelGGL((rowtarget_mse_kernel<LPR, VPL>), dim3(nb), dim3(256), 0, s, z, ld_z, tm, d4, row_idx, coef, \
                     cnt, kind, n_rows, dz, ld_dz, partials)
  switch (lpr) {
    case 1: GD_RT_CASE(1, 1); break;
    case 2: GD_RT_CASE(2, 1); break;
    case 4: GD_RT_CASE(4, 1); break;
    case 8: GD_RT_CASE(8, 1); break;
    case 16: GD_RT_CASE(16, 1); break;
    case 32: GD_RT_CASE(32, 1); break;
    default:
      if (d4 <= 64) GD_RT_CASE(64, 1);
      else if (d4 <= 128) GD_RT_CASE(64, 2);
      else GD_RT_CASE(64, 4);
  }
#undef GD_RT_CASE
  int rc = launched("rowtarget_mse");
  if (rc || !sums) return rc;          // sums == NULL: the caller reduces the partials (gd_loss_finalize_f32)
  hipLaunchKernelGGL(pair_sum_reduce_kernel, dim3(1), dim3(256), 0, s, partials, nb, sums);
  return launched("pair_sum_reduce");
}

extern "C" int32_t gd_rowtarget_mse_pair_covers(int32_t d_a, int32_t d_b) {
  return (d_a == 128 || d_a == 64) && (d_b == 128 || d_b == 64) ? 1 : 0;
}

extern "C" int gd_rowtarget_mse_pair_f32(const float* z_a, int64_t ld_z_a, const float* tm_a, int32_t d_a, const int32_t* row_idx_a,
                                         const float* coef_a, const float* cnt_a, const int32_t* kind_a, int32_t n_rows_a, float* dz_a,
                                         int64_t ld_dz_a, float* partials_a, const float* z_b, int64_t ld_z_b, const float* tm_b,
                                         int32_t d_b, const int32_t* row_idx_b, const float* coef_b, const float* cnt_b,
                                         const int32_t* kind_b, int32_t n_rows_b, float* dz_b, int64_t ld_dz_b, float* partials_b,
                                         void* stream) {
  using namespace gd;
  GD_REQUIRE(gd_rowtarget_mse_pair_covers(d_a, d_b), GD_E_DIM, "gd_rowtarget_mse_pair_f32: widths %d / %d (64 or 128 each; else two gd_rowtarget_mse_f32 calls)", d_a, d_b);
  GD_REQUIRE(n_rows_a > 0 && n_rows_b > 0, GD_E_DIM, "gd_rowtarget_mse_pair_f32: both jobs need rows");
  GD_REQUIRE(z_a && tm_a && row_idx_a && coef_a && cnt_a && kind_a && partials_a && z_b && tm_b && row_idx_b && coef_b && cnt_b && kind_b && partials_b,
             GD_E_NULL, "gd_rowtarget_mse_pair_f32: null pointer");
  GD_REQUIRE(ld_z_a % 4 == 0 && ld_z_b % 4 == 0 && (!dz_a || ld_dz_a % 4 == 0) && (!dz_b || ld_dz_b % 4 == 0), GD_E_DIM,
             "gd_rowtarget_mse_pair_f32: row strides must be multiples of 4");
  GD_REQUIRE(aligned16(z_a) && aligned16(tm_a) && aligned16(dz_a) && aligned16(z_b) && aligned16(tm_b) && aligned16(dz_b), GD_E_ALIGN,
             "gd_rowtarget_mse_pair_f32: unaligned matrix");
  const RowTargetJob a{z_a, ld_z_a, tm_a, d_a / 4, row_idx_a, coef_a, cnt_a, kind_a, n_rows_a, dz_a, ld_dz_a, partials_a};
  const RowTargetJob b{z_b, ld_z_b, tm_b, d_b / 4, row_idx_b, coef_b, cnt_b, kind_b, n_rows_b, dz_b, ld_dz_b, partials_b};
  const int nb_a = (n_rows_a + 255) / 256, nb_b = (n_rows_b + 255) / 256;
  hipStream_t s = (hipStream_t)stream;
#define GD_RTP(LA, LB) hipLaunchKernelGGL((rowtarget_mse_pair_kernel<LA, LB>), dim3(nb_a + nb_b), dim3(256), 0, s, a, b, nb_a)
  if (d_a == 128 && d_b == 64) GD_RTP(32, 16);
  else if (d_a == 128) GD_RTP(32, 32);
  else if (d_b == 64) GD_RTP(16, 16);
  else GD_RTP(16, 32);
#undef GD_RTP
  return launched("rowtarget_mse_pair");
}

extern "C" int gd_edge_dot_f32(const float* z, int64_t ld_z, int32_t d, const int64_t* e0, const int64_t* e1,
                               const float* rel, int64_t ld_rel, const int64_t* etype, int64_t n_edges, float* out,
                               void* stream) {
  using namespace gd;
  if (n_edges == 0) return GD_OK;
  GD_REQUIRE(z && e0 && e1 && out, GD_E_NULL, "gd_edge_dot_f32: null pointer");
  GD_REQUIRE((rel == nullptr) == (etype == nullptr), GD_E_NULL, "gd_edge_dot_f32: rel and etype go together");
  GD_REQUIRE(d > 0 && ld_z >= d, GD_E_DIM, "gd_edge_dot_f32: bad dims");
  hipStream_t s = (hipStream_t)stream;
  const bool vec_ok = d % 4 == 0 && d <= 256 && ld_z % 4 == 0 && aligned16(z) && (!rel || (aligned16(rel) && ld_rel % 4 == 0));
  if (!vec_ok) {
    const dim3 grid((unsigned)((n_edges + 3) / 4));
    if (rel) hipLaunchKernelGGL((edge_dot_scalar_kernel<true>), grid, dim3(256), 0, s, z, ld_z, d, e0, e1, rel, ld_rel, etype, n_edges, out);
    else hipLaunchKernelGGL((edge_dot_scalar_kernel<false>), grid, dim3(256), 0, s, z, ld_z, d, e0, e1, rel, ld_rel, etype, n_edges, out);
    return launched("edge_dot_scalar");
  }
  const int d4 = d / 4;
  const int lpr = lanes_per_row(d4);
  const int64_t per_block = 4 * (kWave / lpr);
  const dim3 grid((unsigned)((n_edges + per_block - 1) / per_block));
#define GD_DOT_CASE(LPR)                                                                                              \
  do {                                                                                                                \
    if (rel) hipLaunchKernelGGL((edge_dot_kernel<LPR, 1, true>), grid, dim3(256), 0, s, z, ld_z, d4, e0, e1, rel, ld_rel, etype, n_edges, out); \
    else hipLaunchKernelGGL((edge_dot_kernel<LPR, 1, false>), grid, dim3(256), 0, s, z, ld_z, d4, e0, e1, rel, ld_rel, etype, n_edges, out);    \
  } while (0)
  switch (lpr) {
    case 1: GD_DOT_CASE(1); break;
    case 2: GD_DOT_CASE(2); break;
    case 4: GD_DOT_CASE(4); break;
    case 8: GD_DOT_CASE(8); break;
    case 16: GD_DOT_CASE(16); break;
    case 32: GD_DOT_CASE(32); break;
    default: GD_DOT_CASE(64); break;
  }
#undef GD_DOT_CASE
  return launched("edge_dot");
}

extern "C" int32_t gd_rowtarget_mse_blocks(int32_t n_rows) { return (n_rows + 255) / 256; }

extern "C" int gd_loss_finalize_f32(const float* partials1, int32_t n1, const float* partials2, int32_t n2,
                                    const float* extra_sums, float* hist, int32_t capacity, int32_t* pos,
                                    int32_t* iter, void* stream) {
  using namespace gd;
  GD_REQUIRE(hist && pos && iter && capacity > 0, GD_E_NULL, "gd_loss_finalize_f32: null pointer");
  GD_REQUIRE((n1 == 0 || partials1) && (n2 == 0 || partials2), GD_E_NULL, "gd_loss_finalize_f32: null partials");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials1, n1, partials2, n2,
                     extra_sums, hist, capacity, pos, iter);
  return launched("loss_finalize");
}

extern "C" int gd_adam_at_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const int32_t* iter,
                              int64_t n, double lr, double beta1, double beta2, double eps, void* stream) {
  using namespace gd;
  GD_REQUIRE(param && grad && exp_avg && exp_avg_sq && iter, GD_E_NULL, "gd_adam_at_f32: null pointer");
  if (n <= 0) return GD_OK;
  hipLaunchKernelGGL(adam_at_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param,
                     grad, exp_avg, exp_avg_sq, iter, n, lr, beta1, beta2, eps);
  return launched("adam_at");
}

extern "C" int gd_adam_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t* step,
                           int64_t n, double lr, double beta1, double beta2, double eps, void* stream) {
  using namespace gd;
  GD_REQUIRE(param && grad && exp_avg && exp_avg_sq && step, GD_E_NULL, "gd_adam_f32: null pointer");
  GD_REQUIRE(n >= 0, GD_E_DIM, "gd_adam_f32: n < 0");
  hipStream_t s = (hipStream_t)stream;
  if (n > 0) {
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, param, grad, exp_avg,
                       exp_avg_sq, step, n, lr, beta1, beta2, eps);
    int rc = launched("adam");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(1), 0, s, step);
  return launched("adam_bump");
}

extern "C" int gd_edge_dot_bwd_f32(const float* z, int64_t ld_z, int32_t d, const int32_t* other, const float* w,
                                   const float* rel, int64_t ld_rel, const int32_t* etype, const int64_t* inc_ptr,
                                   int64_t n_nodes, float* dz, int64_t ld_dz, void* stream) {
  using namespace gd;
  GD_REQUIRE(z && inc_ptr && dz && (!rel || etype), GD_E_NULL, "gd_edge_dot_bwd_f32: null pointer");
  GD_REQUIRE(d > 0 && d % 4 == 0 && d <= 1024 && ld_z % 4 == 0 && ld_dz % 4 == 0 && ld_z >= d && ld_dz >= d && (!rel || ld_rel % 4 == 0),
             GD_E_DIM, "gd_edge_dot_bwd_f32: d=%d must be a multiple of 4 (<= 1024) with 16-byte row strides", d);
  GD_REQUIRE(aligned16(z) && aligned16(dz) && (!rel || aligned16(rel)) && z != dz, GD_E_ALIGN, "gd_edge_dot_bwd_f32: unaligned or aliasing pointer");
  if (n_nodes <= 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const int d4 = d / 4;
  const int lpr = lanes_per_row(d4);
  const int64_t per_block = 4 * (kWave / lpr);
  const dim3 grid((unsigned)((n_nodes + per_block - 1) / per_block));
#define GD_DOTB_CASE(LPR)                                                                                              \
  do {                                                                                                                 \
    if (rel) hipLaunchKernelGGL((edge_dot_bwd_kernel<LPR, true>), grid, dim3(256), 0, s, z, ld_z, d4, other, w, rel, ld_rel, etype, inc_ptr, n_nodes, dz, ld_dz); \
    else hipLaunchKernelGGL((edge_dot_bwd_kernel<LPR, false>), grid, dim3(256), 0, s, z, ld_z, d4, other, w, rel, ld_rel, etype, inc_ptr, n_nodes, dz, ld_dz);    \
  } while (0)
  switch (lpr) {
    case 1: GD_DOTB_CASE(1); break;
    case 2: GD_DOTB_CASE(2); break;
    case 4: GD_DOTB_CASE(4); break;
    case 8: GD_DOTB_CASE(8); break;
    case 16: GD_DOTB_CASE(16); break;
    case 32: GD_DOTB_CASE(32); break;
    default: GD_DOTB_CASE(64); break;
  }
#undef GD_DOTB_CASE
  return launched("edge_dot_bwd");
}

// ---------------------------------------------------------------------------------------------------------------
// The non-MSE row losses of the reference's loss zoo (framework/trainer/gnndelete_nodeemb.py:18-28: BoundedKLD*,
// CosineDistance*), value and gradient with respect to the FIRST argument in one pass over the row pairs:
//   kind 0 (cosine): val[r] = 1 - <a_r, b_r> / (max(|a_r|, 1e-8) max(|b_r|, 1e-8)),   grad[r,:] = d val[r] / d a_r
//   kind 1 (KLD):    val[r] = sum_j t_j (log t_j - log_softmax(a_r)_j), t = softmax(b_r),
//                    grad[r,:] = softmax(a_r) - t      (= d val[r] / d a_r)
// The reductions over the rows (mean / sum, 1 - exp(-KL)) are scalar work left to the caller.  One wave per row, the
// row strided over the lanes (d <= 1024), wave-xor reductions; rows of both operands optionally gathered (ia / ib).
namespace gd {

template <int KIND>
__global__ __launch_bounds__(256) void rowpair_loss_kernel(const float* __restrict__ a, int64_t ld_a, const int64_t* __restrict__ ia,
                                                           const float* __restrict__ b, int64_t ld_b, const int64_t* __restrict__ ib,
                                                           int32_t n_rows, int32_t d, float* __restrict__ val,
                                                           float* __restrict__ grad, int64_t ld_g) {
  constexpr int kMax = 16;                                // d <= 1024
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const float* ar = a + (ia ? ia[r] : (int64_t)r) * ld_a;
  const float* br = b + (ib ? ib[r] : (int64_t)r) * ld_b;
  float av[kMax], bv[kMax];
#pragma unroll
  for (int q = 0; q < kMax; ++q) {
    const int k = lane + q * kWave;
    av[q] = k < d ? ar[k] : (KIND == 1 ? -INFINITY : 0.f);
    bv[q] = k < d ? br[k] : (KIND == 1 ? -INFINITY : 0.f);
  }
  float* gr = grad + (int64_t)r * ld_g;
  if (KIND == 0) {
    float ab = 0.f, aa = 0.f, bb = 0.f;
#pragma unroll
    for (int q = 0; q < kMax; ++q) { ab = fmaf(av[q], bv[q], ab); aa = fmaf(av[q], av[q], aa); bb = fmaf(bv[q], bv[q], bb); }
    ab = wave_sum(ab); aa = wave_sum(aa); bb = wave_sum(bb);
    // F.cosine_similarity clamps each norm at eps = 1e-8: cos = <a, b> / (max(|a|, eps) max(|b|, eps))
    const float na = sqrtf(aa), nb = sqrtf(bb);
    const bool clamped = na < 1e-8f;
    const float den = fmaxf(na, 1e-8f) * fmaxf(nb, 1e-8f);
    const float cs = ab / den;
    if (lane == 0) val[r] = 1.f - cs;
    // d/da (ab / (|a| |b|)) = b / den - cos a / aa   (a clamped |a| is a constant: only b / den remains)
    const float ca = clamped ? 0.f : cs / aa;
#pragma unroll
    for (int q = 0; q < kMax; ++q) {
      const int k = lane + q * kWave;
      if (k < d) gr[k] = -(bv[q] / den - ca * av[q]);
    }
  } else {
    float ma = -INFINITY, mb = -INFINITY;
#pragma unroll
    for (int q = 0; q < kMax; ++q) { ma = fmaxf(ma, av[q]); mb = fmaxf(mb, bv[q]); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { ma = fmaxf(ma, __shfl_xor(ma, off)); mb = fmaxf(mb, __shfl_xor(mb, off)); }
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int q = 0; q < kMax; ++q) { sa += expf(av[q] - ma); sb += expf(bv[q] - mb); }       // exp(-inf) = 0 beyond d
    sa = wave_sum(sa); sb = wave_sum(sb);
    const float lsa = ma + logf(sa), lsb = mb + logf(sb);
    float kl = 0.f;
#pragma unroll
    for (int q = 0; q < kMax; ++q) {
      const int k = lane + q * kWave;
      if (k < d) {
        const float lt = bv[q] - lsb, t = expf(lt), la = av[q] - lsa;
        kl += t > 0.f ? t * (lt - la) : 0.f;                 // (xlogy convention of F.kl_div: 0 where t = 0)
        gr[k] = expf(la) - t;
      }
    }
    kl = wave_sum(kl);
    if (lane == 0) val[r] = kl;
  }
}

}  // namespace gd

extern "C" int gd_rowpair_loss_f32(int32_t kind, const float* a, int64_t ld_a, const int64_t* ia, const float* b, int64_t ld_b,
                                   const int64_t* ib, int32_t n_rows, int32_t d, float* val, float* grad, int64_t ld_g,
                                   void* stream) {
  using namespace gd;
  GD_REQUIRE(a && b && val && grad, GD_E_NULL, "gd_rowpair_loss_f32: null pointer");
  GD_REQUIRE((kind == 0 || kind == 1) && n_rows >= 0 && d > 0 && d <= 1024 && ld_a >= d && ld_b >= d && ld_g >= d, GD_E_DIM,
             "gd_rowpair_loss_f32: kind must be 0 (cosine) or 1 (KLD), d in [1, 1024] (kind=%d d=%d)", kind, d);
  if (n_rows == 0) return GD_OK;
  const dim3 grid((n_rows + 3) / 4), block(256);
  if (kind == 0)
    hipLaunchKernelGGL((rowpair_loss_kernel<0>), grid, block, 0, (hipStream_t)stream, a, ld_a, ia, b, ld_b, ib, n_rows, d, val, grad, ld_g);
  else
    hipLaunchKernelGGL((rowpair_loss_kernel<1>), grid, block, 0, (hipStream_t)stream, a, ld_a, ia, b, ld_b, ib, n_rows, d, val, grad, ld_g);
  return launched("rowpair_loss");
}
