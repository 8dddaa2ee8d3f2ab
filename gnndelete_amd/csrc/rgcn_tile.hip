// R-GCN typed message passing, tile form (PyG RGCNConv aggr='mean' with constant relation weights,
// framework/models/rgcn.py:16-38; same contract as gd_rgcn_conv_f32 in rgcn.hip):
//
//     y[i,:] += sum_r ( sum_{e in run(i,r)} w_e x[col_e,:] ) @ W_r
//
// The node-major kernel (rgcn.hip) re-reads the 16 KB relation weight for EVERY (node, relation) run: 3.3 M runs x
// 16 KB = 55 GB of L2 reads per launch at ogbl-biokg size - it runs at L2 bandwidth (7.6 ms).  Here the work is
// regrouped so that a weight is fetched once per (64-node tile, relation):
//
//   block  = one tile of 64 consecutive nodes, 8 waves: wave (rt, ot) owns the accumulators of the 32 nodes
//            rt x the d_out / 4 outputs ot (one weight block of the reference's num_blocks = 4 configuration) in 16
//            MFMA accumulator registers FOR THE WHOLE TILE - no [R, N, d] tensor, no atomics, no scatter;
//            relations are added in ascending order (deterministic);
//   step   = one relation present in the tile (a second, third ... step for runs longer than 16 edges): the waves
//            first build the step's A tile in LDS - row = node, the weighted sum of that node's run, zero when the
//            node has no edge of this relation - one lane group (d_in / 4 lanes x float4) per run, runs as int2
//            {first edge, row | length << 8} "pieces", edges stored in (tile, relation, node) order;
//   then     D[32 out][32 node] += W_r^T[out][k] A^T[k][32 node] on v_mfma_f32_32x32x2_f32: only the [KL x OW] block of
//            the block-diagonal weight that feeds the wave's outputs (KL = 32 = one block's inputs; a 16-wide block
//            fills half of the MFMA's output rows), pre-packed in the lane order of the MFMA A operand
//            (gd_rgcn_pack_weight_f32: one coalesced 16-byte load per 4 k).
//
// The A tile is double buffered: while MFMA(s) reads one buffer the same waves gather step s + 1 into the other -
// one barrier per step; piece descriptors are fetched two steps ahead and edge (col, w) one step ahead, so a step's
// critical path is one memory round trip (the neighbour rows).  Padding rows without a run to zero costs matrix
// work (a node holds 36 of the 102 relations: 35 % of the rows are live) - which is what keeps the accumulators in
// registers: 78 GF per launch at biokg size, < 1 ms on the matrix cores, against 4.3 GB of gathered rows.
#include "common.h"

namespace gd {

using f32x16t = __attribute__((ext_vector_type(16))) float;

template <int DIN, int OW, int KL>
__global__ __launch_bounds__(512, 4) void rgcn_tile_kernel(
    const int32_t* __restrict__ tile_order, const int32_t* __restrict__ tile_step_ptr, const int32_t* __restrict__ step_rel,
    const int32_t* __restrict__ step_piece_ptr, const uint64_t* __restrict__ step_mask, const int2* __restrict__ piece,
    const int32_t* __restrict__ col, const float* __restrict__ w, const float* __restrict__ x, int64_t ldx,
    const float* __restrict__ wpk, int32_t k0_stride, float* __restrict__ y, int64_t ldy, int32_t n_nodes) {
  constexpr int NW = 8, NT = NW * 64, PITCH = DIN + 4, LPR = DIN / 4, GPW = 64 / LPR, NG = NW * GPW, MAXR = 64 / NG;
  constexpr int J8 = KL / 8;
  extern __shared__ __attribute__((aligned(16))) float a_lds_raw[];   // two A tiles of 64 rows x PITCH floats
  auto a_lds = [&](int b) -> float* { return a_lds_raw + b * (64 * PITCH); };
  const int tile = tile_order ? tile_order[blockIdx.x] : blockIdx.x;
  const int s0 = tile_step_ptr[tile], s1 = tile_step_ptr[tile + 1];
  if (s0 == s1) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rt = wave >> 2, ot = wave & 3, n_lo = lane & 31, khalf = lane >> 5;
  const int grp = wave * GPW + lane / LPR, gl = lane % LPR;

  for (int i = tid; i < 2 * 64 * PITCH / 4; i += NT) reinterpret_cast<float4*>(a_lds_raw)[i] = f4_zero();

  auto load_desc = [&](int s, int2* d) {
    int p0 = 0, p1 = 0;
    if (s < s1) { p0 = step_piece_ptr[s]; p1 = step_piece_ptr[s + 1]; }
#pragma unroll
    for (int j = 0; j < MAXR; ++j) {
      const int pi = p0 + grp + NG * j;
      d[j] = pi < p1 ? piece[pi] : make_int2(0, 0);
    }
  };
  auto load_cw = [&](const int2* d, int* c, float* ww) {
#pragma unroll
    for (int j = 0; j < MAXR; ++j) {
      const bool ok = gl < (d[j].y >> 8);
      c[j] = ok ? col[d[j].x + gl] : 0;
      ww[j] = ok ? w[d[j].x + gl] : 0.f;
    }
  };
  // weighted sums of step s's runs -> buf (rows of `dirty` that this step leaves empty are cleared)
  auto gather = [&](int s, const int2* d, const int* c, const float* ww, float* buf, uint64_t dirty) -> uint64_t {
    const uint64_t mask = step_mask[s];
    const uint64_t zm = dirty & ~mask;
#pragma unroll
    for (int j = 0; j < MAXR; ++j)
      if ((zm >> (grp + NG * j)) & 1) *reinterpret_cast<float4*>(buf + (grp + NG * j) * PITCH + 4 * gl) = f4_zero();
#pragma unroll
    for (int j = 0; j < MAXR; ++j) {
      const int len = d[j].y >> 8, row = d[j].y & 255;
      if (len == 0) continue;
      float4 acc = f4_zero();
      for (int k = 0; k < len; k += 4) {
        float4 v[4];
        float we[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int cc = __shfl(c[j], k + u, LPR);
          we[u] = __shfl(ww[j], k + u, LPR);
          v[u] = k + u < len ? *reinterpret_cast<const float4*>(x + (int64_t)cc * ldx + 4 * gl) : f4_zero();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = f4_fma(we[u], v[u], acc);
      }
      *reinterpret_cast<float4*>(buf + row * PITCH + 4 * gl) = acc;
    }
    return mask;
  };

  f32x16t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  int2 d0[MAXR], d1[MAXR], d2[MAXR];
  int c0[MAXR], c1[MAXR];
  float w0[MAXR], w1[MAXR];
  load_desc(s0, d0);
  load_desc(s0 + 1, d1);
  load_cw(d0, c0, w0);
  __syncthreads();                                        // zero fill done
  uint64_t dirty[2] = {0, 0};
  dirty[0] = gather(s0, d0, c0, w0, a_lds(0), 0);
  load_cw(d1, c1, w1);
  load_desc(s0 + 2, d2);
  __syncthreads();
  int cur = 0;
  for (int s = s0; s < s1; ++s) {
    // this step's weight slice (arrives behind the gathers of the next step)
    constexpr bool kPrefetchW = J8 <= 4;                 // a dense 128-wide slice (64 registers) is read in the loop instead
    const float4* wp = reinterpret_cast<const float4*>(wpk) + ((int64_t)(step_rel[s] * 4 + ot) * J8) * 64 + lane;
    float4 wv[kPrefetchW ? J8 : 1];
    if (kPrefetchW) {
#pragma unroll
      for (int jj = 0; jj < J8; ++jj) wv[jj] = wp[jj * 64];
    }
    if (s + 1 < s1) {
#pragma unroll
      for (int j = 0; j < MAXR; ++j) { d0[j] = d1[j]; c0[j] = c1[j]; w0[j] = w1[j]; d1[j] = d2[j]; }
      load_cw(d1, c1, w1);                                // edges of step s + 2
      load_desc(s + 3, d2);
      dirty[cur ^ 1] = gather(s + 1, d0, c0, w0, a_lds(cur ^ 1), dirty[cur ^ 1]);
    }
    const float* bsrc = a_lds(cur) + (rt * 32 + n_lo) * PITCH + ot * k0_stride + 4 * khalf;
#pragma unroll
    for (int jj = 0; jj < J8; ++jj) {
      const float4 bv = *reinterpret_cast<const float4*>(bsrc + 8 * jj);
      const float4 wj = kPrefetchW ? wv[kPrefetchW ? jj : 0] : wp[jj * 64];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wj.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wj.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wj.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wj.w, bv.w, acc, 0, 0, 0);
    }
    __syncthreads();
    cur ^= 1;
  }
  // D[i][j]: j = lane & 31 = node, output OW ot + 8 q + 4 khalf + c in acc[4 q + c] (rows >= OW are padding)
  const int node = tile * 64 + rt * 32 + n_lo;
  if (node < n_nodes) {
    float* dst = y + (int64_t)node * ldy + OW * ot + 4 * khalf;
#pragma unroll
    for (int q = 0; q < OW / 8; ++q) {
      float4 v = *reinterpret_cast<float4*>(dst + 8 * q);
      v.x += acc[4 * q]; v.y += acc[4 * q + 1]; v.z += acc[4 * q + 2]; v.w += acc[4 * q + 3];
      *reinterpret_cast<float4*>(dst + 8 * q) = v;
    }
  }
}

// packed[((r * 4 + t) * (kl / 8) + jj) * 64 + lane][c] = Wdir_r[t k0_stride + 8 jj + 4 (lane >> 5) + c][ow t + (lane & 31)]
// (0 for lane & 31 >= ow) with Wdir_r the [agg width x out width] block-diagonal matrix of this direction
// (forward: W_r, trans: W_r^T) and ow = d_out / 4 the outputs of one wave
__global__ __launch_bounds__(256) void rgcn_pack_weight_kernel(const float* __restrict__ weight, int32_t n_rel, int32_t n_blocks,
                                                               int32_t ib, int32_t ob, int32_t trans, int32_t ow, int32_t kl,
                                                               int32_t k0_stride, float* __restrict__ packed) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)n_rel * 4 * (kl / 8) * 64 * 4;
  if (e >= total) return;
  const int c = (int)(e & 3), lane = (int)((e >> 2) & 63);
  int64_t rest = e >> 8;
  const int jj = (int)(rest % (kl / 8)); rest /= (kl / 8);
  const int t = (int)(rest & 3), r = (int)(rest >> 2);
  const int k = t * k0_stride + 8 * jj + 4 * (lane >> 5) + c, o = ow * t + (lane & 31);
  const int kb = trans ? ob : ib, nb = trans ? ib : ob;     // agg features / outputs per block in this direction
  float v = 0.f;
  if ((lane & 31) < ow && k / kb == o / nb) {
    const int b = o / nb;
    const float* wb = weight + ((int64_t)r * n_blocks + b) * ib * ob;
    v = trans ? wb[(o % nb) * ob + (k % kb)] : wb[(k % kb) * ob + (o % nb)];
  }
  packed[e] = v;
}

// The k range a wave's d_out / 4 outputs read: one block's inputs for the reference's 4-block weights, everything for
// a dense weight.  (ib, ob) = the FORWARD block.
static bool tile_geometry(int32_t d_in, int32_t d_out, int32_t n_blocks, int32_t ib, int32_t ob, int32_t trans, int* kl, int* k0s) {
  const int kb = trans ? ob : ib, nb = trans ? ib : ob;
  if ((d_in != 64 && d_in != 128) || (d_out != 64 && d_out != 128)) return false;
  if (kb * n_blocks != d_in || nb * n_blocks != d_out) return false;
  if (n_blocks == 4) { *kl = kb; *k0s = kb; return true; }
  if (n_blocks == 1) { *kl = d_in; *k0s = 0; return true; }
  return false;
}

}  // namespace gd

extern "C" int32_t gd_rgcn_tile_kl(int32_t d_in, int32_t d_out, int32_t n_blocks, int32_t trans) {
  if (n_blocks < 1) return 0;
  const int din_f = trans ? d_out : d_in, dout_f = trans ? d_in : d_out;
  if (din_f % n_blocks || dout_f % n_blocks) return 0;
  int kl, k0s;
  return gd::tile_geometry(d_in, d_out, n_blocks, din_f / n_blocks, dout_f / n_blocks, trans, &kl, &k0s) ? kl : 0;
}

extern "C" int gd_rgcn_pack_weight_f32(const float* weight, int32_t n_rel, int32_t n_blocks, int32_t d_in, int32_t d_out,
                                       int32_t trans, float* packed, void* stream) {
  using namespace gd;
  GD_REQUIRE(weight && packed, GD_E_NULL, "gd_rgcn_pack_weight_f32: null pointer");
  const int din_f = trans ? d_out : d_in, dout_f = trans ? d_in : d_out;
  int kl = 0, k0s = 0;
  GD_REQUIRE(n_blocks >= 1 && din_f % n_blocks == 0 && dout_f % n_blocks == 0 &&
                 tile_geometry(d_in, d_out, n_blocks, din_f / n_blocks, dout_f / n_blocks, trans, &kl, &k0s),
             GD_E_DIM, "gd_rgcn_pack_weight_f32: widths / block structure not supported by the tile kernel (d_in=%d d_out=%d blocks=%d)",
             d_in, d_out, n_blocks);
  const int64_t total = (int64_t)n_rel * 4 * (kl / 8) * 256;
  if (total == 0) return GD_OK;
  hipLaunchKernelGGL(rgcn_pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, n_rel,
                     n_blocks, din_f / n_blocks, dout_f / n_blocks, trans, d_out / 4, kl, k0s, packed);
  return launched("rgcn_pack_weight");
}

extern "C" int gd_rgcn_tile_conv_f32(const int32_t* tile_order, const int32_t* tile_step_ptr, const int32_t* step_rel,
                                     const int32_t* step_piece_ptr, const int64_t* step_mask, const int32_t* piece,
                                     const int32_t* col, const float* w, int32_t n_tiles, const float* x, int64_t ldx,
                                     int32_t d_in, const float* packed_w, int32_t n_blocks, int32_t trans, float* y, int64_t ldy,
                                     int32_t d_out, int32_t n_nodes, void* stream) {
  using namespace gd;
  GD_REQUIRE(tile_step_ptr && step_rel && step_piece_ptr && step_mask && piece && col && w && x && packed_w && y, GD_E_NULL,
             "gd_rgcn_tile_conv_f32: null pointer");
  const int din_f = trans ? d_out : d_in, dout_f = trans ? d_in : d_out;
  int kl = 0, k0s = 0;
  GD_REQUIRE(n_blocks >= 1 && din_f % n_blocks == 0 && dout_f % n_blocks == 0 &&
                 tile_geometry(d_in, d_out, n_blocks, din_f / n_blocks, dout_f / n_blocks, trans, &kl, &k0s),
             GD_E_DIM, "gd_rgcn_tile_conv_f32: widths / block structure not supported (d_in=%d d_out=%d blocks=%d); use gd_rgcn_conv_f32",
             d_in, d_out, n_blocks);
  GD_REQUIRE(n_tiles == (n_nodes + 63) / 64 && ldx >= d_in && ldy >= d_out && ldx % 4 == 0 && ldy % 4 == 0, GD_E_DIM,
             "gd_rgcn_tile_conv_f32: n_tiles must be ceil(n_nodes / 64), row pitches multiples of 4");
  GD_REQUIRE(aligned16(x) && aligned16(y) && aligned16(packed_w) && x != y, GD_E_ALIGN, "gd_rgcn_tile_conv_f32: unaligned or aliasing pointer");
  if (n_tiles == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(n_tiles);
  // two A tiles: 67.6 KB at d_in = 128 - above the 64 KB a launch gets without asking
#define GD_RT_CASE(DIN, OW, KL)                                                                                               \
  do {                                                                                                                        \
    constexpr int kLds = 2 * 64 * (DIN + 4) * 4;                                                                              \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&rgcn_tile_kernel<DIN, OW, KL>),        \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds);                      \
    if (attr != hipSuccess) return fail(-(int)attr, "gd_rgcn_tile_conv_f32: %s", hipGetErrorString(attr));                    \
    hipLaunchKernelGGL((rgcn_tile_kernel<DIN, OW, KL>), grid, dim3(512), kLds, s, tile_order, tile_step_ptr, step_rel,  \
                       step_piece_ptr, reinterpret_cast<const uint64_t*>(step_mask), reinterpret_cast<const int2*>(piece), col, \
                       w, x, ldx, packed_w, k0s, y, ldy, n_nodes);                                                             \
  } while (0)
  const int key = d_in * 1000000 + d_out * 1000 + kl;
  switch (key) {
    case 128128032: GD_RT_CASE(128, 32, 32); break;
    case 128128128: GD_RT_CASE(128, 32, 128); break;
    case 128064032: GD_RT_CASE(128, 16, 32); break;
    case 128064128: GD_RT_CASE(128, 16, 128); break;
    case 64128016: GD_RT_CASE(64, 32, 16); break;
    case 64128064: GD_RT_CASE(64, 32, 64); break;
    case 64064016: GD_RT_CASE(64, 16, 16); break;
    case 64064064: GD_RT_CASE(64, 16, 64); break;
    default: return fail(GD_E_DIM, "gd_rgcn_tile_conv_f32: no kernel for d_in=%d d_out=%d kl=%d", d_in, d_out, kl);
  }
#undef GD_RT_CASE
  return launched("rgcn_tile_conv");
}
