// Segment softmax over CSR rows and per-pair row dot products: the attention of the reference's RGATConv
// (framework/models/rgat.py:188-241: additive attention logits per edge, `softmax(alpha, index, ptr, size_i)` across the
// relations of a target node, :322-337 messages alpha_e x_j W_r) as kernels instead of index_reduce / index_add chains.
//
//   segment_softmax fwd   alpha[k] = exp(e[k] - max_row) / (sum_row exp(e - max_row) + 1e-16)       (PyG utils.softmax)
//                   bwd   de[k] = alpha[k] (dalpha[k] - sum_row alpha dalpha)
//   rowpair_dot           out[k] = < a[ia[k], :], b[ib[k], :] >      (d alpha_e = < dm[run(e)], x[src(e)] >)
//
//   typed_wgrad           dW[r, b] = sum over the edges e of relation r of  w_e x[src_e, block b]^T dy[dst_e, block b]
//   typed_edge_dot        out[e] = < x[src_e, :], dy[dst_e, :] W_(r_e)^T >   (block-diagonal or dense relation weights)
//                         (the gradients of TRAINABLE relation weights and of per-edge attention coefficients: the reference
//                          trains them through torch.einsum over an [R, N, d] tensor of per-relation aggregates,
//                          rgcn.py:17-38 under base.py:394-493, rgat.py:188-206 / :322-337 - here straight from the edge lists)
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


// dW[r, b, p, q] for one (relation, diagonal block, 16 x 16 output tile) per workgroup: the relation's edges (a contiguous range
// of the relation-major edge arrays) are walked 64 at a time - their weighted x slices and dy slices staged in LDS - and every
// thread owns one output element and adds the edges in order (fixed order: bit-reproducible, no atomics).
__global__ __launch_bounds__(256) void typed_wgrad_kernel(const int32_t* __restrict__ rel_ptr, const int32_t* __restrict__ src,
                                                          const int32_t* __restrict__ dst, const float* __restrict__ w,
                                                          const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                          int64_t ldy, int32_t n_blocks, int32_t ib, int32_t ob,
                                                          float* __restrict__ dw) {
  __shared__ float xs[64][17], ds[64][17];
  const int tiles_q = (ob + 15) / 16, tiles_p = (ib + 15) / 16;
  int t = blockIdx.x;
  const int tq = t % tiles_q; t /= tiles_q;
  const int tp = t % tiles_p; t /= tiles_p;
  const int b = t % n_blocks, r = t / n_blocks;
  const int p = threadIdx.x >> 4, q = threadIdx.x & 15;
  const int pp = tp * 16 + p, qq = tq * 16 + q;
  const int e0 = rel_ptr[r], e1 = rel_ptr[r + 1];
  float acc = 0.f;
  for (int base = e0; base < e1; base += 64) {
    const int cnt = min(64, e1 - base);
    // stage: thread (edge = tid >> 2, four consecutive columns = 4 (tid & 3))
    const int le = threadIdx.x >> 2, c4 = 4 * (threadIdx.x & 3);
    if (le < cnt) {
      const int e = base + le;
      const float we = w ? w[e] : 1.0f;
      const float* xr = x + (int64_t)src[e] * ldx + b * ib + tp * 16;
      const float* dr = dy + (int64_t)dst[e] * ldy + b * ob + tq * 16;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        xs[le][c4 + c] = (tp * 16 + c4 + c < ib) ? we * xr[c4 + c] : 0.f;
        ds[le][c4 + c] = (tq * 16 + c4 + c < ob) ? dr[c4 + c] : 0.f;
      }
    }
    __syncthreads();
    for (int k = 0; k < cnt; ++k) acc = fmaf(xs[k][p], ds[k][q], acc);
    __syncthreads();
  }
  if (pp < ib && qq < ob) dw[(((int64_t)r * n_blocks + b) * ib + pp) * ob + qq] = acc;
}

// out[e] = sum_b sum_p x[src_e, b ib + p] * (sum_q W[r_e, b, p, q] dy[dst_e, b ob + q]): one wave per edge, lane l owns the input
// features l, l + 64, ...; the lanes' partial products meet in a wave sum (fixed order).
__global__ __launch_bounds__(256) void typed_edge_dot_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                             const int32_t* __restrict__ rel, int64_t n_edges,
                                                             const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                             int64_t ldy, const float* __restrict__ weight, int32_t n_blocks,
                                                             int32_t ib, int32_t ob, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= n_edges) return;
  const float* xr = x + (int64_t)src[e] * ldx;
  const float* dr = dy + (int64_t)dst[e] * ldy;
  const float* wr = weight + (int64_t)rel[e] * n_blocks * ib * ob;
  float acc = 0.f;
  for (int c = lane; c < n_blocks * ib; c += kWave) {
    const int b = c / ib, p = c % ib;
    const float* wp = wr + ((int64_t)b * ib + p) * ob;
    float tq = 0.f;
    for (int q = 0; q < ob; ++q) tq = fmaf(wp[q], dr[b * ob + q], tq);
    acc = fmaf(xr[c], tq, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) out[e] = acc;
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

extern "C" int gd_typed_wgrad_f32(const int32_t* rel_ptr, int32_t n_rel, const int32_t* src, const int32_t* dst, const float* w,
                                  const float* x, int64_t ldx, const float* dy, int64_t ldy, int32_t n_blocks, int32_t d_in,
                                  int32_t d_out, float* dw, void* stream) {
  using namespace gd;
  GD_REQUIRE(rel_ptr && x && dy && dw, GD_E_NULL, "gd_typed_wgrad_f32: null pointer");
  GD_REQUIRE(n_rel >= 0 && n_blocks >= 1 && d_in > 0 && d_out > 0 && d_in % n_blocks == 0 && d_out % n_blocks == 0 && ldx >= d_in &&
                 ldy >= d_out, GD_E_DIM, "gd_typed_wgrad_f32: bad dims (d_in=%d d_out=%d blocks=%d)", d_in, d_out, n_blocks);
  if (n_rel == 0) return GD_OK;
  const int ib = d_in / n_blocks, ob = d_out / n_blocks;
  const int64_t nblk = (int64_t)n_rel * n_blocks * ((ib + 15) / 16) * ((ob + 15) / 16);
  GD_REQUIRE(nblk < (1ll << 31), GD_E_DIM, "gd_typed_wgrad_f32: too many output tiles");
  hipLaunchKernelGGL(typed_wgrad_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, rel_ptr, src, dst, w, x, ldx, dy, ldy,
                     n_blocks, ib, ob, dw);
  return launched("typed_wgrad");
}

extern "C" int gd_typed_edge_dot_f32(const int32_t* src, const int32_t* dst, const int32_t* rel, int64_t n_edges, const float* x,
                                     int64_t ldx, const float* dy, int64_t ldy, const float* weight, int32_t n_blocks, int32_t d_in,
                                     int32_t d_out, float* out, void* stream) {
  using namespace gd;
  GD_REQUIRE(n_edges == 0 || (src && dst && rel && x && dy && weight && out), GD_E_NULL, "gd_typed_edge_dot_f32: null pointer");
  GD_REQUIRE(n_edges >= 0 && n_blocks >= 1 && d_in > 0 && d_out > 0 && d_in % n_blocks == 0 && d_out % n_blocks == 0 && ldx >= d_in &&
                 ldy >= d_out, GD_E_DIM, "gd_typed_edge_dot_f32: bad dims (d_in=%d d_out=%d blocks=%d)", d_in, d_out, n_blocks);
  if (n_edges == 0) return GD_OK;
  GD_REQUIRE((n_edges + 3) / 4 < (1ll << 31), GD_E_DIM, "gd_typed_edge_dot_f32: too many edges");
  hipLaunchKernelGGL(typed_edge_dot_kernel, dim3((unsigned)((n_edges + 3) / 4)), dim3(256), 0, (hipStream_t)stream, src, dst, rel, n_edges,
                     x, ldx, dy, ldy, weight, n_blocks, d_in / n_blocks, d_out / n_blocks, out);
  return launched("typed_edge_dot");
}
