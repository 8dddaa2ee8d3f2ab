// R-GCN typed message passing, tile form (PyG RGCNConv aggr='mean' with constant relation weights,
// framework/models/rgcn.py:16-38; same contract as gd_rgcn_conv_f32 in rgcn.hip):
//
//     y[i,:] += sum_r ( sum_{e in run(i,r)} w_e x[col_e,:] ) @ W_r
//
// The node-major kernel (rgcn.hip) re-reads the 16 KB relation weight for EVERY (node, relation) run: 3.3 M runs x
// 16 KB = 55 GB of L2 reads per launch at ogbl-biokg size - it runs at L2 bandwidth (7.6 ms).  Here the work is
// regrouped so that a weight is fetched once per (64-node tile, relation):
//
//   block  = one tile of 64 consecutive nodes, 8 waves; the tile's outputs [64 x d_out] live in LDS for the whole tile -
//            no [R, N, d] tensor, no atomics; relations are added in ascending order (deterministic);
//   step   = one relation present in the tile (further steps for runs longer than 16 edges and beyond 32 runs): the waves
//            first build the step's COMPACT A tile in LDS - row q = the weighted sum of the step's q-th run - one lane
//            group (d_in / 4 lanes x float4) per run, runs as int2 {first edge, row | length << 8} "pieces", edges
//            stored in (tile, relation, node) order; a row map remembers which node each compact row belongs to;
//   then     D[out][run] += W_r^T[out][k] A^T[k][run] on v_mfma_f32_16x16x4_f32: wave (ot, ch) multiplies the [KL x OW]
//            block of the block-diagonal weight that feeds its d_out / 4 outputs with the runs [16 ch, 16 ch + 16) and
//            adds its 16 x 16 products to the runs' node rows of the accumulators (distinct rows within a step,
//            disjoint (ot, ch) ranges: plain read-modify-write).  The weight block is pre-packed in the lane order of
//            the MFMA A operand (gd_rgcn_pack_weight_f32: one coalesced 16-byte load per 4 k).
//
// Compact rows instead of one row per node: a node holds 36 of the 102 relations, so node-aligned rows would leave 65 %
// of the MFMA columns empty (the first version of this kernel: 0.55 ms of matrix time per launch at biokg size);
// steps with <= 16 runs keep half of the waves out of the product altogether.
//
// Hubs: "one piece per node and step" serialises a node whose runs hold thousands of edges (ogbl-biokg's largest
// entity: 10,423 in-edges, 1,895 of one relation = 119 passes).  The plan spreads such a node over up to 64 SLICE rows
// of extra tiles (piece k of a run -> slice k mod V); a slice row is an ordinary row to this kernel except that its
// outputs go to y_ext, and rgcn_hub_fixup_kernel adds a hub's slices to its row of y in slice order (deterministic).
//
// One A tile per block, two barriers per step (products of step s | gather of step s + 1): at 53 KB of LDS three blocks
// share a CU, and the third block hides more of the gathers' latency than a second A buffer did inside a block
// (measured: 1.25 -> 1.21 ms per launch).  Step scalars sit in an LDS ring, piece descriptors and edge (col, w) pairs
// are fetched into registers a step ahead of their use.
#include "common.h"

namespace gd {

using f32x16t = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using f32x4t = __attribute__((ext_vector_type(4))) float;
constexpr int kRing = 256;                              // steps in the LDS ring (power of two)

template <int DIN, int OW, int KL>
__global__ __launch_bounds__(512, 6) void rgcn_tile_kernel(
    const int32_t* __restrict__ tile_order, const int32_t* __restrict__ tile_step_ptr, const int32_t* __restrict__ step_rel,
    const int32_t* __restrict__ step_piece_ptr, const int2* __restrict__ piece,
    const int32_t* __restrict__ col, const float* __restrict__ w, const float* __restrict__ x, int64_t ldx,
    const float* __restrict__ wpk, int32_t k0_stride, float* __restrict__ y, int64_t ldy, int32_t n_nodes,
    float* __restrict__ y_ext, int32_t n_pad, int64_t n_x_bytes) {
  constexpr int NW = 8, NT = NW * 64, PITCH = DIN + 4, LPR = DIN / 4, GPW = 64 / LPR;
  constexpr int PPW = 4, MAXR = PPW / GPW;                 // pieces per wave and step (a step holds <= 32), rounds to sum them
  constexpr int DOUT = 4 * OW, OPITCH = DOUT + 4, NOH = OW / 16, NMM = KL / 16;
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  // LDS: two compact A tiles (32 pieces x PITCH), the tile's accumulators (64 nodes x OPITCH), the step ring, row maps
  // ONE compact A tile: with 53 KB per block three blocks share a CU (24 waves) - the kernel waits on memory more than on
  // anything else, and the second buffer bought less overlap inside a block than a third block buys across blocks
  float* const a_tile = lds_raw;
  float* const acc_lds = lds_raw + 32 * PITCH;
  int32_t* const m_rel = reinterpret_cast<int32_t*>(acc_lds + 64 * OPITCH);
  int32_t* const m_pp = m_rel + kRing;
  int32_t* const rowmap = m_pp + kRing;                     // [32]: node row of the piece in compact row q
  const int tile = tile_order ? tile_order[blockIdx.x] : blockIdx.x;
  const int s0 = tile_step_ptr[tile], s1 = tile_step_ptr[tile + 1];
  if (s0 == s1) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ot = wave & 3, ch = wave >> 2;                  // MFMA role: outputs [OW ot, OW ot + OW), pieces [16 ch, 16 ch + 16)
  const int g_in_wave = lane / LPR, gl = lane % LPR;

  for (int i = tid; i < 64 * OPITCH / 4; i += NT) reinterpret_cast<float4*>(acc_lds)[i] = f4_zero();
  auto ring_fill = [&](int first, int count) {            // steps [first, first + count) (+ the piece pointer one past the end)
    for (int i = tid; i < count; i += NT) {
      const int s = first + i;
      if (s <= s1) {
        m_pp[(s - s0) & (kRing - 1)] = step_piece_ptr[s];
        if (s < s1) m_rel[(s - s0) & (kRing - 1)] = step_rel[s];
      }
    }
  };
  ring_fill(s0, 192);
  __syncthreads();

  // Plan data per wave and step, one element per lane: the wave takes the pieces wave, wave + 8, wave + 16, wave + 24 of
  // a step; lane i < 4 holds the descriptor of its i-th piece, lane 8 i + t the edges t and t + 8 of that piece.
  struct Cw { int ca, cb; float wa, wb; };
  auto load_desc = [&](int s) -> int2 {
    int2 d = make_int2(0, 0);
    if (s < s1 && lane < PPW) {
      const int p0 = m_pp[(s - s0) & (kRing - 1)], p1 = m_pp[(s + 1 - s0) & (kRing - 1)];
      const int pi = p0 + wave + 8 * lane;
      if (pi < p1) d = piece[pi];
    }
    return d;
  };
  auto load_cw = [&](int2 d) -> Cw {
    const int slot = (lane >> 3) & (PPW - 1);
    const int e0 = __shfl(d.x, slot), len = lane < 8 * PPW ? __shfl(d.y, slot) >> 8 : 0, t = lane & 7;
    Cw r = {0, 0, 0.f, 0.f};
    if (t < len) { r.ca = col[e0 + t]; r.wa = w[e0 + t]; }
    if (t + 8 < len) { r.cb = col[e0 + 8 + t]; r.wb = w[e0 + 8 + t]; }
    return r;
  };
  // neighbour rows through a raw buffer descriptor: a slot beyond the run gets an out-of-range offset, for which the
  // hardware returns zeros without touching memory (no branch around the load), 24 x 24-bit row offsets
  const uint32_t row_bytes = (uint32_t)(ldx * 4);
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (uint32_t)n_x_bytes, 0x00020000);
  auto load_row = [&](uint32_t cc, bool valid) -> float4 {
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, valid ? __umul24(cc, row_bytes) + 16u * gl : 0xffffffffu, 0, 0);
    return __builtin_bit_cast(float4, r);
  };
  // weighted sums of a step's runs -> the compact rows of buf (row q = the step's q-th piece; rows beyond the step's
  // piece count keep stale data whose products are never added).  Round j: lane group g of the wave sums the wave's
  // piece GPW j + g.  Every cross-lane read sits in wave-uniform control flow (ds_bpermute returns 0 for a masked lane).
  auto gather = [&](int2 d, Cw cw, float* buf, int32_t* rmap) {
#pragma unroll
    for (int j = 0; j < MAXR; ++j) {
      int rl = 0, kmax = 0;
#pragma unroll
      for (int g = 0; g < GPW; ++g) {
        const int v = __builtin_amdgcn_readlane(d.y, GPW * j + g);
        kmax = max(kmax, v >> 8);
        rl = g_in_wave == g ? v : rl;
      }
      if (kmax == 0) continue;
      const int len = rl >> 8, slot = GPW * j + g_in_wave, q = wave + 8 * slot, src0 = 8 * slot;
      float4 acc = f4_zero();
      for (int k = 0; k < kmax; k += 4) {
        float4 v[4];
        float we[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int src = src0 + ((k + u) & 7);
          const int cc = k < 8 ? __shfl(cw.ca, src) : __shfl(cw.cb, src);
          we[u] = k < 8 ? __shfl(cw.wa, src) : __shfl(cw.wb, src);
          v[u] = load_row((uint32_t)cc, k + u < len);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = f4_fma(we[u], v[u], acc);
      }
      if (len > 0) {
        *reinterpret_cast<float4*>(buf + q * PITCH + 4 * gl) = acc;
        if (gl == 0) rmap[q] = rl & 255;
      }
    }
  };

  int2 d_cur = load_desc(s0), d_nxt = load_desc(s0 + 1);
  Cw c_cur = load_cw(d_cur);
  gather(d_cur, c_cur, a_tile, rowmap);
  d_cur = d_nxt;
  c_cur = load_cw(d_cur);
  d_nxt = load_desc(s0 + 2);
  __syncthreads();
  const float4* const wpk4 = reinterpret_cast<const float4*>(wpk);
  for (int s = s0; s < s1; ++s) {
    const int n = m_pp[(s + 1 - s0) & (kRing - 1)] - m_pp[(s - s0) & (kRing - 1)];     // pieces of this step (<= 32)
    const bool mm_on = 16 * ch < n;                        // the second half of the waves only works on crowded steps
    // this step's weight block in MFMA lane order (arrives behind the gathers of the next step)
    float4 wv[NOH][NMM];
    if (mm_on) {
      const float4* wp = wpk4 + ((int64_t)(m_rel[(s - s0) & (kRing - 1)] * 4 + ot) * (NOH * NMM)) * 64 + lane;
#pragma unroll
      for (int oh = 0; oh < NOH; ++oh)
#pragma unroll
        for (int mm = 0; mm < NMM; ++mm) wv[oh][mm] = wp[(oh * NMM + mm) * 64];
    }
    if (mm_on) {
      // D[16 out][16 piece] += W^T[out][k] A^T[k][piece] on v_mfma_f32_16x16x4_f32: lane (j = lane & 15, kq = lane >> 4)
      // feeds k = k0 + 16 mm + 4 kq + c of piece 16 ch + j (one 16-byte LDS read per mm) and ends with the outputs
      // 16 oh + 4 kq + c of that piece - added to the piece's node row of the accumulators (distinct rows per step,
      // the (ot, ch) ranges of the waves are disjoint: no atomics, relations in ascending order)
      const int j = lane & 15, kq = lane >> 4;
      const float* bsrc = a_tile + (16 * ch + j) * PITCH + ot * k0_stride + 4 * kq;
      f32x4t dacc[NOH];
#pragma unroll
      for (int oh = 0; oh < NOH; ++oh) dacc[oh] = f32x4t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mm = 0; mm < NMM; ++mm) {
        const float4 bv = *reinterpret_cast<const float4*>(bsrc + 16 * mm);
#pragma unroll
        for (int oh = 0; oh < NOH; ++oh) {
          dacc[oh] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[oh][mm].x, bv.x, dacc[oh], 0, 0, 0);
          dacc[oh] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[oh][mm].y, bv.y, dacc[oh], 0, 0, 0);
          dacc[oh] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[oh][mm].z, bv.z, dacc[oh], 0, 0, 0);
          dacc[oh] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[oh][mm].w, bv.w, dacc[oh], 0, 0, 0);
        }
      }
      if (16 * ch + j < n) {
        float* dst = acc_lds + rowmap[16 * ch + j] * OPITCH + OW * ot + 4 * kq;
#pragma unroll
        for (int oh = 0; oh < NOH; ++oh) {
          float4 v = *reinterpret_cast<float4*>(dst + 16 * oh);
          v.x += dacc[oh][0]; v.y += dacc[oh][1]; v.z += dacc[oh][2]; v.w += dacc[oh][3];
          *reinterpret_cast<float4*>(dst + 16 * oh) = v;
        }
      }
    }
    __syncthreads();                                       // every wave is done with the A tile and the row map
    if (s + 1 < s1) gather(d_cur, c_cur, a_tile, rowmap);
    d_cur = d_nxt;
    c_cur = load_cw(d_cur);
    d_nxt = load_desc(s + 3);
    __syncthreads();                                       // the next step's tile is complete
    if (((s + 1 - s0) & 63) == 0) ring_fill(s + 1 + 128, 64);   // slots of steps every wave has left behind
  }
  // the tile's rows: y += accumulators (a hub's slice rows go to y_ext, added up by the fix-up)
  for (int i = tid; i < 64 * (DOUT / 4); i += NT) {
    const int r = i / (DOUT / 4), c4 = i % (DOUT / 4), node = tile * 64 + r;
    const float4 a = *reinterpret_cast<const float4*>(acc_lds + r * OPITCH + 4 * c4);
    if (node < n_nodes) {
      float4* dst = reinterpret_cast<float4*>(y + (int64_t)node * ldy) + c4;
      *dst = f4_add(*dst, a);
    } else if (node >= n_pad) {
      reinterpret_cast<float4*>(y_ext + (int64_t)(node - n_pad) * DOUT)[c4] = a;
    }
  }
}

// y[hub_node[h], :] += the hub's slice rows y_ext[hub_ptr[h] .. hub_ptr[h + 1]), in slice order
__global__ __launch_bounds__(128) void rgcn_hub_fixup_kernel(const int32_t* __restrict__ hub_node, const int32_t* __restrict__ hub_ptr,
                                                             const float* __restrict__ y_ext, int32_t d_out,
                                                             float* __restrict__ y, int64_t ldy) {
  const int h = blockIdx.x, f = threadIdx.x;
  if (f >= d_out) return;
  float acc = 0.f;
  for (int v = hub_ptr[h]; v < hub_ptr[h + 1]; ++v) acc += y_ext[(int64_t)v * d_out + f];
  y[(int64_t)hub_node[h] * ldy + f] += acc;
}

// packed[(((r * 4 + t) * (ow / 16) + oh) * (kl / 16) + mm) * 64 + lane][c] =
//     Wdir_r[t k0_stride + 16 mm + 4 (lane >> 4) + c][ow t + 16 oh + (lane & 15)]
// with Wdir_r the [agg width x out width] block-diagonal matrix of this direction (forward: W_r, trans: W_r^T) and
// ow = d_out / 4 the outputs of one wave: the A operand of v_mfma_f32_16x16x4_f32, one 16-byte load per four k steps
__global__ __launch_bounds__(256) void rgcn_pack_weight_kernel(const float* __restrict__ weight, int32_t n_rel, int32_t n_blocks,
                                                               int32_t ib, int32_t ob, int32_t trans, int32_t ow, int32_t kl,
                                                               int32_t k0_stride, float* __restrict__ packed) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int noh = ow / 16, nmm = kl / 16;
  const int64_t total = (int64_t)n_rel * 4 * noh * nmm * 64 * 4;
  if (e >= total) return;
  const int c = (int)(e & 3), lane = (int)((e >> 2) & 63);
  int64_t rest = e >> 8;
  const int mm = (int)(rest % nmm); rest /= nmm;
  const int oh = (int)(rest % noh); rest /= noh;
  const int t = (int)(rest & 3), r = (int)(rest >> 2);
  const int k = t * k0_stride + 16 * mm + 4 * (lane >> 4) + c, o = ow * t + 16 * oh + (lane & 15);
  const int kb = trans ? ob : ib, nb = trans ? ib : ob;     // agg features / outputs per block in this direction
  float v = 0.f;
  if (k / kb == o / nb) {
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
  const int64_t total = (int64_t)n_rel * 4 * ((d_out / 4) / 16) * (kl / 16) * 256;
  if (total == 0) return GD_OK;
  hipLaunchKernelGGL(rgcn_pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, n_rel,
                     n_blocks, din_f / n_blocks, dout_f / n_blocks, trans, d_out / 4, kl, k0s, packed);
  return launched("rgcn_pack_weight");
}

extern "C" int gd_rgcn_tile_conv_f32(const int32_t* tile_order, const int32_t* tile_step_ptr, const int32_t* step_rel,
                                     const int32_t* step_piece_ptr, const int32_t* piece,
                                     const int32_t* col, const float* w, int32_t n_tiles, const float* x, int64_t ldx,
                                     int32_t d_in, const float* packed_w, int32_t n_blocks, int32_t trans, float* y, int64_t ldy,
                                     int32_t d_out, int32_t n_nodes, const int32_t* hub_node, const int32_t* hub_ptr,
                                     int32_t n_hubs, float* y_ext, void* stream) {
  using namespace gd;
  GD_REQUIRE(tile_step_ptr && step_rel && step_piece_ptr && piece && col && w && x && packed_w && y, GD_E_NULL,
             "gd_rgcn_tile_conv_f32: null pointer");
  const int din_f = trans ? d_out : d_in, dout_f = trans ? d_in : d_out;
  int kl = 0, k0s = 0;
  GD_REQUIRE(n_blocks >= 1 && din_f % n_blocks == 0 && dout_f % n_blocks == 0 &&
                 tile_geometry(d_in, d_out, n_blocks, din_f / n_blocks, dout_f / n_blocks, trans, &kl, &k0s),
             GD_E_DIM, "gd_rgcn_tile_conv_f32: widths / block structure not supported (d_in=%d d_out=%d blocks=%d); use gd_rgcn_conv_f32",
             d_in, d_out, n_blocks);
  const int n_real = (n_nodes + 63) / 64;
  GD_REQUIRE(n_tiles >= n_real && ldx >= d_in && ldy >= d_out && ldx % 4 == 0 && ldy % 4 == 0 && n_hubs >= 0, GD_E_DIM,
             "gd_rgcn_tile_conv_f32: n_tiles must be at least ceil(n_nodes / 64), row pitches multiples of 4");
  GD_REQUIRE(n_tiles == n_real || (n_hubs > 0 && hub_node && hub_ptr && y_ext && aligned16(y_ext)), GD_E_NULL,
             "gd_rgcn_tile_conv_f32: slice tiles need hub_node / hub_ptr / y_ext");
  GD_REQUIRE(aligned16(x) && aligned16(y) && aligned16(packed_w) && x != y, GD_E_ALIGN, "gd_rgcn_tile_conv_f32: unaligned or aliasing pointer");
  GD_REQUIRE(n_nodes <= (1 << 24) && ldx * 4 < (1 << 24) && (int64_t)n_nodes * ldx * 4 < ((int64_t)1 << 32), GD_E_DIM,
             "gd_rgcn_tile_conv_f32: x beyond 4 GB / 2^24 rows (24 x 24-bit row offsets); use gd_rgcn_conv_f32");
  if (n_tiles == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(n_tiles);
  // two A tiles: 67.6 KB at d_in = 128 - above the 64 KB a launch gets without asking
#define GD_RT_CASE(DIN, OW, KL)                                                                                               \
  do {                                                                                                                        \
    constexpr int kLds = (32 * (DIN + 4) + 64 * (4 * OW + 4)) * 4 + kRing * 8 + 32 * 4;                                                                             \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&rgcn_tile_kernel<DIN, OW, KL>),        \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds);                      \
    if (attr != hipSuccess) return fail(-(int)attr, "gd_rgcn_tile_conv_f32: %s", hipGetErrorString(attr));                    \
    hipLaunchKernelGGL((rgcn_tile_kernel<DIN, OW, KL>), grid, dim3(512), kLds, s, tile_order, tile_step_ptr, step_rel,  \
                       step_piece_ptr, reinterpret_cast<const int2*>(piece), col, \
                       w, x, ldx, packed_w, k0s, y, ldy, n_nodes, y_ext, n_real * 64, ((int64_t)(n_nodes - 1) * ldx + d_in) * 4);                                         \
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
  int rc = launched("rgcn_tile_conv");
  if (rc || n_tiles == n_real) return rc;
  hipLaunchKernelGGL(rgcn_hub_fixup_kernel, dim3(n_hubs), dim3(128), 0, s, hub_node, hub_ptr, y_ext, d_out, y, ldy);
  return launched("rgcn_hub_fixup");
}
