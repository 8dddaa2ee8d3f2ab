// Last-layer Del operator, its loss and its input gradient in one pass over the S_Df rows:
//
//     z  = p[idx,:] @ W_D                         DeletionLayer.forward (deletion.py:17-29)
//     dz = coef_u (z - tbar_u)                    folded DEC + NI MSE terms of the layer (loss.hip), 0 without a slot
//     dp[idx,:] = dz @ W_D^T                      what autograd sends back through the Del operator
//
// z itself has no other consumer during training (it is the model output), so it is never written: per
// row the kernel reads p and tbar and writes dz (for the weight gradient) and dp (for conv2's backward) -
// 4 row-streams instead of the 10 of the separate Del / loss / Del-backward kernels.
//
// Same mapping as rows_gemm.hip (transposed product, weight images in LDS, sample rows straight from global
// memory); W_D is held twice, once per product, each in the order its MFMA operands are consumed, so that one
// vector read (8 / 16 bytes) feeds NT / 4 consecutive MFMAs (one 4-byte read per MFMA before);
// every LDS address is one per-lane base plus a compile-time offset.  The second product needs no layout change: after the first one a lane
// holds, for its own sample j, the features i = 32t + (r&3) + 8(r>>2) + 4kh in acc[t][r]; using exactly that
// feature as MFMA k slot (r, kh) of the second product - both operands agree on the permutation - the
// accumulator registers ARE its "B" operand, and its "A" operand W_D[c][i] comes out of the same LDS image
// read along the other axis.
//
// WG form (gd_del_loss_bwd_wgrad_f32): the Del weight's gradient  dW_D = p[idx,:]^T dz  in the same pass - the third product
// of the tile, K = its 32 samples.  Both operands are in this wave's registers already, but sample-major (a lane owns a
// sample); the product wants them feature-major (lane = feature, K slot = sample), so each goes once through a wave-private
// 32 x (D + 4) LDS tile: p right after its fetch (its transposed view is read back into the registers the row-major copy
// leaves after the forward product), dz over the same tile after the loss.  The wave keeps the D x D sums of all its tiles in
// registers (D^2 / 64 per lane); at the end the block adds its four waves' sums in wave order and writes ONE partial matrix,
// laid out and counted like gd_rows_gemm_wgrad_f32's (grid = gd_rows_gemm_wgrad_blocks(n_sel)), so the same reduction
// (gd_step_tail_f32 / gd_rows_gemm_wgrad_reduce_f32) finishes it.  No dz buffer, no second pass over p: the separate
// weight-gradient launch read 97 MB and took 26 us of the bench step.  To make room for the sums the targets are no
// longer fetched a tile ahead but at the top of their own tile (the forward product's 2 us cover them).
#include <stdlib.h>

#include "common.h"

namespace gd {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct DelLoss {
  const int32_t* slot;      // [n_sel] loss slot of selected row s, or -1
  const float* tm;          // [n_slots, d]
  const float* coef;        // [n_slots]
  const float* cnt_signed;  // [n_slots] (NI slots negative)
  float* partials;          // [2 * gridDim.x]
};

// (the 64-wide form WITH the weight-gradient sums needs 270 registers: one wave per SIMD for that instantiation instead of 56 B of
//  scratch per lane at two - it only runs below 65,536 rows, where the launch is latency-sized; the step's form is the kernel below)
template <int NT, bool WG>
__global__ __launch_bounds__(256, (NT == 2 && WG) ? 1 : 2) void del_loss_bwd_kernel(const float* __restrict__ p, int64_t ld_p,
                                                              const int32_t* __restrict__ idx, int32_t n_sel,
                                                              const float* __restrict__ w, DelLoss loss,
                                                              float* __restrict__ dz, int64_t ld_dz,
                                                              float* __restrict__ dp, int64_t ld_dp,
                                                              float* __restrict__ wg_partials) {
  constexpr int D = 32 * NT;
  constexpr int PB = D + 4;                                      // pitch of the row-major image (16-byte rows, conflict-free b128)
  constexpr int PT = D + 4;                                      // pitch of the wave's transposition tile (WG)
  // two images of W_D, one per product, so that the NT (4) operands of consecutive MFMAs are ONE vector read:
  //   fw[(k * 32 + n % 32) * NT + n / 32] = W_D[k][n]   forward:  the NT output tiles of one k step
  //   bw[k * PB + n]                      = W_D[k][n]   backward: four consecutive k slots of one output tile
  extern __shared__ __attribute__((aligned(16))) float wl[];
  float* const fw = wl;
  float* const bw = wl + D * D;
  __shared__ float lred[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, kh = lane >> 5;
  for (int e = tid; e < D * D; e += 256) {
    const int k = e / D, n = e % D;
    const float v = w[e];
    fw[(k * 32 + (n & 31)) * NT + (n >> 5)] = v;
    bw[k * PB + n] = v;
  }
  __syncthreads();

  const float* w_fwd = fw + (kh * (D / 2) * 32 + lo) * NT;        // per-lane bases of the two read patterns
  const float* w_bwd = bw + lo * PB + 4 * kh;
  float* const tb = wl + D * D + D * PB + wave * (32 * PT);       // (WG) this wave's 32-sample tile
  float* const tb_row = tb + lo * PT + 4 * kh;                    // sample-major side: my sample's row
  const float* const tb_col = tb + kh * PT + lo;                  // feature-major side: K slot (st, kh) = sample 2 st + kh
  f32x16 gacc[WG ? NT * NT : 1];                                  // (WG) dW_D[32 mt + (r&3) + 8(r>>2) + 4kh][32 nt + lo] in gacc[mt * NT + nt][r]
  if (WG) {
#pragma unroll
    for (int i = 0; i < NT * NT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) gacc[i][r] = 0.f;
  }
  float ls0 = 0.f, ls1 = 0.f;
  const int n_tiles = (n_sel + 31) >> 5;
  // Operands of one tile: sample row index, loss slot, this lane's half of the p row, its target runs.
  // The next tile's operands are requested before the current tile's 2 x NT x D/2 MFMAs are issued
  // (plain arrays, copied element-wise: keeps everything in registers).
  int32_t n_row;
  int n_u;
  bool n_live;
  float n_cf, n_cn;
  float4 n_pa[D / 8], n_tv[NT * 4];
  auto fetch = [&](int tile_) {
    const int s_ = tile_ * 32 + lo;
    n_live = s_ < n_sel;
    const int sc = min(s_, n_sel - 1);
    n_row = idx[sc];
    n_u = n_live ? loss.slot[sc] : -1;
    if (!WG) {
      const float4* src = reinterpret_cast<const float4*>(p + (int64_t)n_row * ld_p + kh * (D / 2));
#pragma unroll
      for (int c4 = 0; c4 < D / 8; ++c4) n_pa[c4] = src[c4];
    }
    const int uc = max(n_u, 0);
    n_cf = loss.coef[uc];
    n_cn = loss.cnt_signed[uc];
    if (!WG) {
      const float* trow = loss.tm + (int64_t)uc * D + 4 * kh;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) n_tv[t * 4 + q] = *reinterpret_cast<const float4*>(trow + 32 * t + 8 * q);
    }
  };
  // (WG) the next tile's p rows are asked for only after this tile's weight-gradient product, when the registers of the
  // transposed p operand are free again: the row-major copy, its transposed view, the D x D sums and a prefetched tile do not
  // fit 256 registers together (56 B/lane of scratch when they were tried to); the Del-backward product covers the fetch
  auto fetch_rows = [&]() {
    const float4* src = reinterpret_cast<const float4*>(p + (int64_t)n_row * ld_p + kh * (D / 2));
#pragma unroll
    for (int c4 = 0; c4 < D / 8; ++c4) n_pa[c4] = src[c4];
  };
  const int stride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  if (tile < n_tiles) {
    fetch(tile);
    if (WG) fetch_rows();
  }
  for (; tile < n_tiles; tile += stride) {
    // the weight fragments a lane reads from LDS are the same for every tile; left alone the compiler hoists
    // all 2 NT D/2 of them out of this loop and spills - keep them as loads inside the loop
    asm volatile("" ::: "memory");
    const int s = tile * 32 + lo;
    const bool live = n_live;
    const int32_t row = n_row;
    const int u = n_u;
    const float c_cf = n_cf, c_cn = n_cn;
    float4 c_pa[D / 8], c_tv[NT * 4];
#pragma unroll
    for (int c4 = 0; c4 < D / 8; ++c4) c_pa[c4] = n_pa[c4];
    if (WG) {                                                     // this tile's targets: asked for now, needed after the forward product
      const float* trow = loss.tm + (int64_t)max(u, 0) * D + 4 * kh;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) c_tv[t * 4 + q] = *reinterpret_cast<const float4*>(trow + 32 * t + 8 * q);
#pragma unroll
      for (int c4 = 0; c4 < D / 8; ++c4) *reinterpret_cast<float4*>(tb_row + kh * (D / 2 - 4) + 4 * c4) = c_pa[c4];   // p, sample-major
    } else {
#pragma unroll
      for (int q = 0; q < NT * 4; ++q) c_tv[q] = n_tv[q];
    }
    if (tile + stride < n_tiles) fetch(tile + stride);
    // ---- z^T tile = W_D^T x^T : lane (sample lo, half kh) feeds features [kh*D/2, kh*D/2 + D/2) of its row
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < D / 8; ++c4) {
      const float4 a4 = c_pa[c4];
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* wk = w_fwd + (4 * c4 + e) * 32 * NT;        // W_D[kh*D/2 + 4c4 + e][lo + 32t], t = 0 .. NT-1 adjacent
        float wv[NT];
        if (NT == 2) { const float2 f = *reinterpret_cast<const float2*>(wk); wv[0] = f.x; wv[NT - 1] = f.y; }
        else wv[0] = wk[0];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t], av[e], acc[t], 0, 0, 0);
      }
    }
    // (WG) the transposed view of p: a_op[st * NT + mt] = p[sample 2 st + kh][feature 32 mt + lo]
    // The tile is wave-private: lanes write sample-major rows and read OTHER lanes' data feature-major.  DS operations of a
    // wave execute in issue order; the wavefront-scope fences keep the compiler from reordering the tile's writes and
    // transposed reads (in either direction, across tiles too) on a disjointness-per-lane argument.
    float a_op[WG ? 16 * NT : 1];
    if (WG) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int st = 0; st < 16; ++st)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) a_op[st * NT + mt] = tb_col[2 * st * PT + 32 * mt];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    // ---- loss gradient in place: acc[t][r] (feature 32t + (r&3) + 8(r>>2) + 4kh of sample lo) -> dz
    if (u >= 0) {
      const float cf = c_cf, cn = c_cn;
      float* drow = dz + (int64_t)s * ld_dz + 4 * kh;
      float sq = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 tv = c_tv[t * 4 + q];
          const float4 df = make_float4(acc[t][4 * q] - tv.x, acc[t][4 * q + 1] - tv.y, acc[t][4 * q + 2] - tv.z,
                                        acc[t][4 * q + 3] - tv.w);
          sq = fmaf(df.x, df.x, sq); sq = fmaf(df.y, df.y, sq); sq = fmaf(df.z, df.z, sq); sq = fmaf(df.w, df.w, sq);
          const float4 g4 = make_float4(cf * df.x, cf * df.y, cf * df.z, cf * df.w);
          acc[t][4 * q] = g4.x; acc[t][4 * q + 1] = g4.y; acc[t][4 * q + 2] = g4.z; acc[t][4 * q + 3] = g4.w;
          if (!WG || dz) *reinterpret_cast<float4*>(drow + 32 * t + 8 * q) = g4;
          if (WG) *reinterpret_cast<float4*>(tb_row + 32 * t + 8 * q) = g4;
        }
      if (cn >= 0.f) ls0 = fmaf(cn, sq, ls0); else ls1 = fmaf(-cn, sq, ls1);
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      if (live && (!WG || dz)) {
        float* drow = dz + (int64_t)s * ld_dz + 4 * kh;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(drow + 32 * t + 8 * q) = f4_zero();
      }
      if (WG) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(tb_row + 32 * t + 8 * q) = f4_zero();
      }
    }
    // ---- (WG) dW_D += p^T dz over the tile's 32 samples: A = p (registers, feature-major), B = dz (the tile, feature-major)
    if (WG) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int st = 0; st < 16; ++st)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float b = tb_col[2 * st * PT + 32 * nt];
#pragma unroll
          for (int mt = 0; mt < NT; ++mt)
            gacc[mt * NT + nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_op[st * NT + mt], b, gacc[mt * NT + nt], 0, 0, 0);
        }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();               // (before the tile is overwritten)
      asm volatile("" ::: "memory");                              // (the compiler would issue the fetch at the top of the tile)
      if (tile + stride < n_tiles) fetch_rows();
    }
    // ---- dp^T tile = W_D dz^T : k slot (kk, kh) of feature tile t <-> feature i = 32t + (kk&3) + 8(kk>>2) + 4kh
    f32x16 dacc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) dacc[c][r] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int c = 0; c < NT; ++c) {                                            // W_D[32c + lo][32t + 8m + 4kh + j], j = 0..3
          const float4 f = *reinterpret_cast<const float4*>(w_bwd + 32 * c * PB + 32 * t + 8 * m);
          dacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, acc[t][4 * m + 0], dacc[c], 0, 0, 0);
          dacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, acc[t][4 * m + 1], dacc[c], 0, 0, 0);
          dacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, acc[t][4 * m + 2], dacc[c], 0, 0, 0);
          dacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, acc[t][4 * m + 3], dacc[c], 0, 0, 0);
        }
    if (live) {
      float* orow = dp + (int64_t)row * ld_dp + 4 * kh;
#pragma unroll
      for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(orow + 32 * c + 8 * q) =
              make_float4(dacc[c][4 * q], dacc[c][4 * q + 1], dacc[c][4 * q + 2], dacc[c][4 * q + 3]);
    }
  }
  if (WG) {
    __syncthreads();                                              // nobody reads the weight images or a tile any more
    float* const red = wl;                                        // [4 waves][D x D]
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          red[wave * D * D + (32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kh) * D + 32 * nt + lo] = gacc[mt * NT + nt][r];
    __syncthreads();
    float* const out = wg_partials + (int64_t)blockIdx.x * D * D;
    for (int e = tid; e < D * D; e += 256) out[e] = (red[e] + red[D * D + e]) + (red[2 * D * D + e] + red[3 * D * D + e]);
  }
  ls0 = wave_sum(ls0);
  ls1 = wave_sum(ls1);
  if (lane == 0) { lred[0][wave] = ls0; lred[1][wave] = ls1; }
  __syncthreads();
  if (tid == 0) {
    loss.partials[2 * blockIdx.x + 0] = (lred[0][0] + lred[0][1]) + (lred[0][2] + lred[0][3]);
    loss.partials[2 * blockIdx.x + 1] = (lred[1][0] + lred[1][1]) + (lred[1][2] + lred[1][3]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight-stationary form of the WG kernel for D = 64 (round 4; same idea as rows_gemm_ws.hip): ONE wave per SIMD keeps W_D
// twice as MFMA A-fragments in its registers - wr1 for z = p W_D, wr2 for dp = dz W_D^T (64 + 64 registers, fetched straight
// from global memory: W_D is 16 KB) - and the 64 x 64 weight-gradient sums of all its rows in 64 more; v_mfma_f32_16x16x4_f32,
// work unit = 16 rows, lane (r = lane & 15, kq = lane >> 4):
//   P1  z[r][16 t + 4 kq + c]  = sum_s  W[kq 16 + s][16 t + .] p[r][kq 16 + s]          (B = the lane's 64-byte slice of its p row)
//   loss: dz = coef_u (z - tbar_u) in the accumulator layout (targets fetched in that layout), loss sums
//   P2  dp[r][16 t2 + 4 kq + c] = sum_(t,c) W[16 t2 + .][16 t + 4 kq + c] dz[r][16 t + 4 kq + c]   (B = P1's accumulators: the k order
//       of the second product is the feature order the first one left in the registers - no shuffle)
//   P3  dW[16 ta + 4 kq + v][16 tb + j] += sum_rows p[row][16 ta + .] dz[row][16 tb + j]: K = the unit's 16 rows; both operands go
//       once through a wave-private 16 x 80 LDS tile to come back feature-major (written before P2, read back under it); the
//       product itself is issued one unit later, under the next unit's loss arithmetic.
// A unit's row / target registers are reloaded with the next unit's as soon as their last use is issued (row ids and loss
// slots two units ahead), dp is stored under P3.  No block barrier in the loop, no scratch (the 32-row form above: 256 VGPRs + 56 B/lane of scratch at two waves per
// SIMD).  Rows past the end are clamped to the last row and recompute it (identical stores); their loss / weight-gradient
// contributions are zeroed.  Partial matrices: block b writes its sum to slot b and zeros to the slots b + grid, ... < n_part
// (the reduction kernels count gd_rows_gemm_wgrad_blocks(n_sel) partials).
typedef float f32x4w __attribute__((ext_vector_type(4)));
template <bool HAS_DZ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void del_loss_bwd_ws_kernel(
    const float* __restrict__ p, int64_t ld_p, const int32_t* __restrict__ idx, int32_t n_sel, const float* __restrict__ w,
    DelLoss loss, float* __restrict__ dz, int64_t ld_dz, float* __restrict__ dp, int64_t ld_dp, float* __restrict__ wg_partials,
    int32_t n_part) {
  constexpr int D = 64, PT = 80;                                  // PT: pitch of the transposition tiles (rows 16 banks apart)
  extern __shared__ __attribute__((aligned(16))) float wl[];      // 4 waves x (2 tiles of 16 x PT), later 4 x D x D block sum
  __shared__ float lred[2][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kq = lane >> 4;
  float* const tp = wl + wave * (2 * 16 * PT);
  float* const tz = tp + 16 * PT;
  const int n_units = (n_sel + 15) >> 4;
  const int n_waves = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
  const int u_lo = (int)((int64_t)n_units * wid / n_waves), u_hi = (int)((int64_t)n_units * (wid + 1) / n_waves);

  auto slot_of = [&](int u) -> int { return min(min(u, n_units - 1) * 16 + r, n_sel - 1); };
  // descriptors two units ahead, rows / targets / coefficients one unit ahead
  int32_t row_n = idx[slot_of(u_lo)], ls_n = loss.slot[slot_of(u_lo)];
  int32_t row_nn = idx[slot_of(u_lo + 1)], ls_nn = loss.slot[slot_of(u_lo + 1)];
  __builtin_amdgcn_sched_barrier(0);
  float wr1[4][16], wr2[4][16];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) wr1[t][s_] = w[(kq * 16 + s_) * D + 16 * t + r];
#pragma unroll
  for (int t2 = 0; t2 < 4; ++t2)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float4 f = *reinterpret_cast<const float4*>(w + (16 * t2 + r) * D + 16 * t + 4 * kq);
      wr2[t2][4 * t + 0] = f.x; wr2[t2][4 * t + 1] = f.y; wr2[t2][4 * t + 2] = f.z; wr2[t2][4 * t + 3] = f.w;
    }
  f32x4w gacc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) gacc[i] = f32x4w{0.f, 0.f, 0.f, 0.f};
  float ls0 = 0.f, ls1 = 0.f;

  // Ring: the row / target / coefficient registers of a unit are reloaded with the NEXT unit's as soon as their last use
  // is issued (P2 + P3, ~2 us, to land); nothing is copied.
  float4 x[4], tv[4];
  float cf_raw, cn_raw;                                            // (masked by the slot's sign where they are USED: nothing may
  int32_t ls_cur;                                                  //  touch a loaded value before the next unit, or its wait lands here)
  auto fetch_targets = [&](int32_t u) {                            // branch-free: slot -1 reads slot 0, coefficient masked at use
    const int uc = max(u, 0);
    const float* trow = loss.tm + (int64_t)uc * D + 4 * kq;
#pragma unroll
    for (int t = 0; t < 4; ++t) tv[t] = *reinterpret_cast<const float4*>(trow + 16 * t);
    cf_raw = loss.coef[uc];
    cn_raw = loss.cnt_signed[uc];
    ls_cur = u;
  };
  auto fetch_rows = [&](int32_t row) {
    const float4* src = reinterpret_cast<const float4*>(p + (int64_t)row * ld_p + kq * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = src[i];
  };
  // feature-major operands of P3, one unit behind: unit u's weight-gradient product is issued under unit u + 1's loss
  // arithmetic and tile writes (the matrix pipe would idle there), the last unit's after the loop; zeros the first time round
  float a_op[4][4], b_op[4][4];
#pragma unroll
  for (int sp = 0; sp < 4; ++sp)
#pragma unroll
    for (int t = 0; t < 4; ++t) a_op[sp][t] = b_op[sp][t] = 0.f;
  auto p3 = [&]() {
#pragma unroll
    for (int sp = 0; sp < 4; ++sp)
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
          gacc[ta * 4 + tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_op[sp][ta], b_op[sp][tb], gacc[ta * 4 + tb], 0, 0, 0);
  };
  if (u_lo < u_hi) {
    fetch_rows(row_n);
    fetch_targets(ls_n);
    for (int u = u_lo; u < u_hi; ++u) {
      const int32_t row = row_n;
      const int s_a = min(u, n_units - 1) * 16 + r;
      const float livef = s_a < n_sel ? 1.f : 0.f;
      row_n = row_nn;
      ls_n = ls_nn;
      row_nn = idx[slot_of(u + 2)];
      ls_nn = loss.slot[slot_of(u + 2)];
      // ---- P1: z
      f32x4w acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xv[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (i == 0 && c == 0) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr1[t][0], xv[0], f32x4w{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr1[t][4 * i + c], xv[c], acc[t], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- P3 of the PREVIOUS unit (operands in registers) - the scheduler interleaves it with what follows
      p3();
      // ---- loss gradient in the accumulator layout; p and dz into the transposition tiles
      const float cf = ls_cur >= 0 ? cf_raw : 0.f, cn = ls_cur >= 0 ? cn_raw : 0.f;
      float sq = 0.f;
      f32x4w g[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 d4 = make_float4(acc[t][0] - tv[t].x, acc[t][1] - tv[t].y, acc[t][2] - tv[t].z, acc[t][3] - tv[t].w);
        sq = fmaf(d4.x, d4.x, sq); sq = fmaf(d4.y, d4.y, sq); sq = fmaf(d4.z, d4.z, sq); sq = fmaf(d4.w, d4.w, sq);
        g[t] = f32x4w{cf * d4.x, cf * d4.y, cf * d4.z, cf * d4.w};
        if (HAS_DZ) *reinterpret_cast<float4*>(dz + (int64_t)min(s_a, n_sel - 1) * ld_dz + 16 * t + 4 * kq) = make_float4(g[t][0], g[t][1], g[t][2], g[t][3]);
        *reinterpret_cast<float4*>(tz + r * PT + 16 * t + 4 * kq) = make_float4(livef * g[t][0], livef * g[t][1], livef * g[t][2], livef * g[t][3]);
      }
      sq *= livef;
      if (cn >= 0.f) ls0 = fmaf(cn, sq, ls0); else ls1 = fmaf(-cn, sq, ls1);
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(tp + r * PT + kq * 16 + 4 * i) = x[i];
      __builtin_amdgcn_sched_barrier(0);
      fetch_targets(ls_n);                                         // the next unit's operands into the registers just consumed
      fetch_rows(row_n);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the tiles are wave-private: DS operations execute in issue order
      __builtin_amdgcn_wave_barrier();
      // feature-major operands of this unit's P3: requested here, they arrive under P2
#pragma unroll
      for (int sp = 0; sp < 4; ++sp)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          a_op[sp][t] = tp[(4 * sp + kq) * PT + 16 * t + r];
          b_op[sp][t] = tz[(4 * sp + kq) * PT + 16 * t + r];
        }
      __builtin_amdgcn_sched_barrier(0);
      // ---- P2: dp = dz W_D^T
      f32x4w dacc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t2 = 0; t2 < 4; ++t2) {
            if (t == 0 && c == 0) dacc[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr2[t2][0], g[0][0], f32x4w{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else dacc[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr2[t2][4 * t + c], g[t][c], dacc[t2], 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
      {
        float* orow = dp + (int64_t)row * ld_dp + 4 * kq;
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) *reinterpret_cast<float4*>(orow + 16 * t2) = make_float4(dacc[t2][0], dacc[t2][1], dacc[t2][2], dacc[t2][3]);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the tiles are read before the next unit overwrites them)
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    p3();                                                          // the last unit's weight-gradient product
  }
  // ---- block sum of the four waves' D x D sums (wave order), loss sums
  __syncthreads();
  float* const red = wl;                                          // [4][D x D]
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[wave * D * D + (16 * ta + 4 * kq + v) * D + 16 * tb + r] = gacc[ta * 4 + tb][v];
  __syncthreads();
  for (int slot = blockIdx.x; slot < n_part; slot += gridDim.x) {
    float* const out = wg_partials + (int64_t)slot * D * D;
    const bool mine = slot == (int)blockIdx.x;
    for (int e = tid; e < D * D; e += 256) out[e] = mine ? (red[e] + red[D * D + e]) + (red[2 * D * D + e] + red[3 * D * D + e]) : 0.f;
  }
  ls0 = wave_sum(ls0);
  ls1 = wave_sum(ls1);
  if (lane == 0) { lred[0][wave] = ls0; lred[1][wave] = ls1; }
  __syncthreads();
  if (tid == 0) {
    for (int slot = blockIdx.x; slot < n_part; slot += gridDim.x) {
      const bool mine = slot == (int)blockIdx.x;
      loss.partials[2 * slot + 0] = mine ? (lred[0][0] + lred[0][1]) + (lred[0][2] + lred[0][3]) : 0.f;
      loss.partials[2 * slot + 1] = mine ? (lred[1][0] + lred[1][1]) + (lred[1][2] + lred[1][3]) : 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// FIRST-layer Del operator at 128 features, its folded loss and its weight gradient in one pass over the S_Df rows (round 5):
//
//     z[idx,:] = p[idx,:] @ W_D   (+ the packed [z > 0] pattern conv2's backward gates with)            DeletionLayer.forward
//     g        = coef_u (z - tbar_u) + g_add[idx,:]     folded DEC + NI terms of the layer + the gradient that arrives through conv2
//     dW_D    += p[idx,:]^T g                           (per-block partial matrices, reduced by gd_step_tail_f32)
//
// What the step ran before: the weight-stationary row GEMM (reads p, writes z: 59 us) and the loss-fused weight-gradient kernel
// (reads p AGAIN, reads z back, reads tbar and g_add: 83 us at 0.45 of the matrix peak) - 549 MB; here p is read once and z is
// only written: 366 MB.  The same weight-stationary scheme as del_loss_bwd_ws_kernel above, with the 128 x 128 weight and the
// 128 x 128 gradient sums split by OUTPUT COLUMN HALF over the two waves of a pair: wave (pair, half) keeps W_D[:, 64 half ..]
// as 128 registers of MFMA A-fragments and dW_D[:, 64 half ..] in 128 more, both waves fetch the pair's 16 rows (the second
// fetch hits the L1 / L2), each forms its half of z, of the loss terms and of g; v_mfma_f32_16x16x4_f32, lane (r = lane & 15,
// kq = lane >> 4):
//   P1  z[r][cb + 16 t + 4 kq + c] = sum_(i,c') W[16 i + 4 kq + c'][cb + 16 t + .] p[r][16 i + 4 kq + c']   (B = the lane's eight float4 of its p row)
//   loss / g in the accumulator layout (targets and g_add fetched in that layout), z stored, sign bits merged over kq
//   P3  dW[16 ta + 4 kq + v][cb + 16 tb + j] += sum_rows p[row][16 ta + .] g[row][cb + 16 tb + j]: K = the unit's 16 rows; p and
//       g go once through wave-private LDS tiles (16 x 144, 16 x 80) to come back feature-major; the product is issued
//       right behind the NEXT unit's fetch, whose latency it covers.
// One wave per SIMD, no block barrier in the loop.  Partial matrices: block b writes its sum to slot b and zeros to the slots
// b + grid, ... < n_part (the reduction counts gd_rows_gemm_wgrad_blocks(n_sel) partials).
template <bool HAS_ADD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void del1_loss_wgrad_ws_kernel(
    const float* __restrict__ p, int64_t ld_p, const int32_t* __restrict__ idx, int32_t n_sel, const float* __restrict__ w,
    float* __restrict__ z, int64_t ld_z, uint32_t* __restrict__ sign_out, DelLoss loss, const float* __restrict__ g_add,
    int64_t ld_ga, float* __restrict__ wg_partials, int32_t n_part) {
  constexpr int D = 128, H = 64, PTP = 144, PTZ = 80;             // pitches: rows 16 banks apart
  extern __shared__ __attribute__((aligned(16))) float wl[];      // 4 waves x (16 x PTP + 16 x PTZ), later 4 x D x H block sums
  __shared__ float lred[2][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave & 1, pair = wave >> 1, cb = H * half;
  const int r = lane & 15, kq = lane >> 4;
  float* const tp = wl + wave * (16 * PTP + 16 * PTZ);
  float* const tz = tp + 16 * PTP;
  const int n_units = (n_sel + 15) >> 4;
  const int n_pairs = gridDim.x * 2, pid = blockIdx.x * 2 + pair;
  const int u_lo = (int)((int64_t)n_units * pid / n_pairs), u_hi = (int)((int64_t)n_units * (pid + 1) / n_pairs);

  auto slot_of = [&](int u) -> int { return min(min(u, n_units - 1) * 16 + r, n_sel - 1); };
  int32_t row_n = idx[slot_of(u_lo)], ls_n = loss.slot[slot_of(u_lo)];
  int32_t row_nn = idx[slot_of(u_lo + 1)], ls_nn = loss.slot[slot_of(u_lo + 1)];
  __builtin_amdgcn_sched_barrier(0);
  float wr1[4][32];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int s_ = 0; s_ < 32; ++s_) wr1[t][s_] = w[(kq * 32 + s_) * D + cb + 16 * t + r];
  f32x4w gacc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) gacc[i] = f32x4w{0.f, 0.f, 0.f, 0.f};
  float ls0 = 0.f, ls1 = 0.f;

  float4 x[8], tv[4], av[4];
  float cf_raw, cn_raw;
  int32_t ls_cur;
  auto fetch_targets = [&](int32_t u, int32_t row) {              // branch-free: slot -1 reads slot 0, coefficient masked at use
    const int uc = max(u, 0);
    const float* trow = loss.tm + (int64_t)uc * D + cb + 4 * kq;
#pragma unroll
    for (int t = 0; t < 4; ++t) tv[t] = *reinterpret_cast<const float4*>(trow + 16 * t);
    if (HAS_ADD) {
      const float* arow = g_add + (int64_t)row * ld_ga + cb + 4 * kq;
#pragma unroll
      for (int t = 0; t < 4; ++t) av[t] = *reinterpret_cast<const float4*>(arow + 16 * t);
    }
    cf_raw = loss.coef[uc];
    cn_raw = loss.cnt_signed[uc];
    ls_cur = u;
  };
  auto fetch_rows = [&](int32_t row) {
    const float4* src = reinterpret_cast<const float4*>(p + (int64_t)row * ld_p + kq * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = src[i];
  };
  if (u_lo < u_hi) {
    fetch_rows(row_n);
    fetch_targets(ls_n, row_n);
    for (int u = u_lo; u < u_hi; ++u) {
      const int32_t row = row_n;
      const int s_a = min(u, n_units - 1) * 16 + r;
      const bool live = s_a < n_sel;
      const float livef = live ? 1.f : 0.f;
      row_n = row_nn;
      ls_n = ls_nn;
      row_nn = idx[slot_of(u + 2)];
      ls_nn = loss.slot[slot_of(u + 2)];
      // ---- P1: this wave's 64 columns of z
      f32x4w acc[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xv[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (i == 0 && c == 0) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr1[t][0], xv[0], f32x4w{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr1[t][4 * i + c], xv[c], acc[t], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- the p rows into their transposition tile (rows past the end zeroed: no weight-gradient contribution)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        *reinterpret_cast<float4*>(tp + r * PTP + kq * 32 + 4 * i) = make_float4(livef * x[i].x, livef * x[i].y, livef * x[i].z, livef * x[i].w);
      // ---- loss gradient in the accumulator layout; z out, sign bits, g into its tile
      const float cf = ls_cur >= 0 ? cf_raw : 0.f, cn = ls_cur >= 0 ? cn_raw : 0.f;
      float sq = 0.f;
      uint32_t bits[2] = {0u, 0u};
      float* zrow = z + (int64_t)row * ld_z + cb + 4 * kq;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 zv = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        *reinterpret_cast<float4*>(zrow + 16 * t) = zv;
        bits[t >> 1] |= ((zv.x > 0.f ? 1u : 0u) | (zv.y > 0.f ? 2u : 0u) | (zv.z > 0.f ? 4u : 0u) | (zv.w > 0.f ? 8u : 0u)) << (16 * (t & 1) + 4 * kq);
        const float4 d4 = make_float4(zv.x - tv[t].x, zv.y - tv[t].y, zv.z - tv[t].z, zv.w - tv[t].w);
        sq = fmaf(d4.x, d4.x, sq); sq = fmaf(d4.y, d4.y, sq); sq = fmaf(d4.z, d4.z, sq); sq = fmaf(d4.w, d4.w, sq);
        float4 gv = make_float4(cf * d4.x, cf * d4.y, cf * d4.z, cf * d4.w);
        if (HAS_ADD) gv = f4_add(gv, av[t]);
        *reinterpret_cast<float4*>(tz + r * PTZ + 16 * t + 4 * kq) = gv;
      }
      sq *= livef;
      if (cn >= 0.f) ls0 = fmaf(cn, sq, ls0); else ls1 = fmaf(-cn, sq, ls1);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bits[q] |= (uint32_t)__shfl_xor((int)bits[q], 16);
        bits[q] |= (uint32_t)__shfl_xor((int)bits[q], 32);
      }
      if (live && kq == 0) {
        sign_out[(int64_t)s_a * 4 + 2 * half + 0] = bits[0];
        sign_out[(int64_t)s_a * 4 + 2 * half + 1] = bits[1];
      }
      __builtin_amdgcn_sched_barrier(0);
      fetch_rows(row_n);                                           // the next unit's operands into the registers just consumed
      fetch_targets(ls_n, row_n);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the tiles are wave-private: DS operations execute in issue order
      __builtin_amdgcn_wave_barrier();
      // ---- P3: this unit's weight-gradient product from the feature-major views (under the fetch just issued)
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        float a_op[8], b_op[4];
#pragma unroll
        for (int ta = 0; ta < 8; ++ta) a_op[ta] = tp[(4 * sp + kq) * PTP + 16 * ta + r];
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) b_op[tb] = tz[(4 * sp + kq) * PTZ + 16 * tb + r];
#pragma unroll
        for (int ta = 0; ta < 8; ++ta)
#pragma unroll
          for (int tb = 0; tb < 4; ++tb)
            gacc[ta * 4 + tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_op[ta], b_op[tb], gacc[ta * 4 + tb], 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the tiles are read before the next unit overwrites them)
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- block sum of the waves' D x H sums (pair order), loss sums
  __syncthreads();
  float* const red = wl;                                          // [4][D x H]
#pragma unroll
  for (int ta = 0; ta < 8; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[wave * D * H + (16 * ta + 4 * kq + v) * H + 16 * tb + r] = gacc[ta * 4 + tb][v];
  __syncthreads();
  for (int slot = blockIdx.x; slot < n_part; slot += gridDim.x) {
    float* const out = wg_partials + (int64_t)slot * D * D;
    const bool mine = slot == (int)blockIdx.x;
    for (int e = tid; e < D * D; e += 256) {
      const int i = e >> 7, c = e & 127, hf = c >> 6, ch = c & 63;
      out[e] = mine ? red[hf * D * H + i * H + ch] + red[(2 + hf) * D * H + i * H + ch] : 0.f;
    }
  }
  ls0 = wave_sum(ls0);
  ls1 = wave_sum(ls1);
  if (lane == 0) { lred[0][wave] = ls0; lred[1][wave] = ls1; }
  __syncthreads();
  if (tid == 0) {
    for (int slot = blockIdx.x; slot < n_part; slot += gridDim.x) {
      const bool mine = slot == (int)blockIdx.x;
      loss.partials[2 * slot + 0] = mine ? (lred[0][0] + lred[0][1]) + (lred[0][2] + lred[0][3]) : 0.f;
      loss.partials[2 * slot + 1] = mine ? (lred[1][0] + lred[1][1]) + (lred[1][2] + lred[1][3]) : 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same pass with the SECOND gradient stream formed in the kernel instead of read (round 5, GCN / GIN layer-wise steps):
//     g_add[row,:] = (dt[row, 0:64] @ W_next) (.) [z_prev[row,:] > 0]     - conv2's input gradient of the PREVIOUS iteration, from
// the aggregated layer-2 gradient dt that iteration left ([N, 64]) and the sign pattern ITS Del-1 pass stored - read here for a row
// before this iteration's pattern of the same row is written by the same wave.  The stand-alone gated product (a [S1, 128] write
// per step and its read back here: 183 MB, 40 us) is gone for 64 more matrix instructions per unit.  Both weights are LDS images
// in the lanes' read order (W_D 64 KB, W_next 32 KB; in registers - 128 + 64 of them - nothing else would fit the 256 VGPRs the
// compiler keeps MFMA A / B operands in); k order of the products: lane (r, kq) holds p[r][16 i + 4 kq + c] in x[i].c and
// dt[r][16 i + 4 kq + c] in dv[i].c.  Otherwise as del1_loss_wgrad_ws_kernel: 16-row units, output columns split by half over the
// two waves of a pair, one wave per SIMD, partial matrices per block.
// RANK1 (GATConv behind the gate): the formed gradient gets ra[row] ua[col] + rb[row] ub[col] added BEFORE the gate - the two
// rank-1 terms of GAT's input gradient (d a_src (x) att_src W2 and d a_dst (x) att_dst W2; gd_rows_gemm_gated_rank1_f32's epilogue).
struct ChainRank1 { const float* ra; const float* ua; const float* rb; const float* ub; };
template <bool RANK1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void del1_chain_ws_kernel(
    const float* __restrict__ p, int64_t ld_p, const int32_t* __restrict__ idx, int32_t n_sel, const float* __restrict__ w,
    float* __restrict__ z, int64_t ld_z, uint32_t* __restrict__ sign_io, DelLoss loss, const float* __restrict__ dt, int64_t ld_dt,
    const float* __restrict__ w_next, float* __restrict__ wg_partials, int32_t n_part, ChainRank1 r1) {
  constexpr int D = 128, H = 64, PTP = 144, PTZ = 80;
  constexpr int kTiles = 4 * (16 * PTP + 16 * PTZ);               // floats of the four waves' transposition tiles
  extern __shared__ __attribute__((aligned(16))) float wl[];      // tiles | W_D image | W_next image; later 4 x D x H block sums
  __shared__ float lred[2][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave & 1, pair = wave >> 1, cb = H * half;
  const int r = lane & 15, kq = lane >> 4;
  float* const tp = wl + wave * (16 * PTP + 16 * PTZ);
  float* const tz = tp + 16 * PTP;
  // wimg[((h 4 + t) 8 + i) 64 + lane] = W_D[16 i + 4 kq + 0..3][64 h + 16 t + r];  w2img[((h 4 + t) 4 + i) 64 + lane] = W_next[...] likewise
  float4* const wimg = reinterpret_cast<float4*>(wl + kTiles);
  float4* const w2img = wimg + 2 * 4 * 8 * 64;
  {
    float4 fv[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int e = tid + 256 * it;
      const int ln = e & 63, i_ = (e >> 6) & 7, t_ = (e >> 9) & 3, h_ = e >> 11;
      const float* src = w + (int64_t)(16 * i_ + 4 * (ln >> 4)) * D + 64 * h_ + 16 * t_ + (ln & 15);
      fv[it] = make_float4(src[0], src[D], src[2 * D], src[3 * D]);
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) wimg[tid + 256 * it] = fv[it];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int e = tid + 256 * it;
      const int ln = e & 63, i_ = (e >> 6) & 3, t_ = (e >> 8) & 3, h_ = e >> 10;
      const float* src = w_next + (int64_t)(16 * i_ + 4 * (ln >> 4)) * D + 64 * h_ + 16 * t_ + (ln & 15);
      fv[it] = make_float4(src[0], src[D], src[2 * D], src[3 * D]);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) w2img[tid + 256 * it] = fv[it];
  }
  __syncthreads();
  const float4* const wmine = wimg + (half * 4) * 8 * 64 + lane;
  const float4* const w2mine = w2img + (half * 4) * 4 * 64 + lane;
  const int n_units = (n_sel + 15) >> 4;
  const int n_pairs = gridDim.x * 2, pid = blockIdx.x * 2 + pair;
  const int u_lo = (int)((int64_t)n_units * pid / n_pairs), u_hi = (int)((int64_t)n_units * (pid + 1) / n_pairs);

  auto slot_of = [&](int u) -> int { return min(min(u, n_units - 1) * 16 + r, n_sel - 1); };
  int32_t row_n = idx[slot_of(u_lo)], ls_n = loss.slot[slot_of(u_lo)];
  int32_t row_nn = idx[slot_of(u_lo + 1)], ls_nn = loss.slot[slot_of(u_lo + 1)];
  __builtin_amdgcn_sched_barrier(0);
  f32x4w gacc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) gacc[i] = f32x4w{0.f, 0.f, 0.f, 0.f};
  float ls0 = 0.f, ls1 = 0.f;

  float4 x[8], dv[4], tv[4];
  float4 ua4[RANK1 ? 4 : 1], ub4[RANK1 ? 4 : 1];                    // (RANK1) this lane's columns of the two column vectors
  float ra_row = 0.f, rb_row = 0.f;                                 // (RANK1) this lane's row scalars
  if (RANK1) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ua4[t] = *reinterpret_cast<const float4*>(r1.ua + cb + 16 * t + 4 * kq);
      ub4[t] = *reinterpret_cast<const float4*>(r1.ub + cb + 16 * t + 4 * kq);
    }
  }
  uint2 old_bits;                                                   // the row's previous sign words of this wave's column half
  float cf_raw, cn_raw;
  int32_t ls_cur;
  auto fetch_targets = [&](int32_t u) {                           // branch-free: slot -1 reads slot 0, coefficient masked at use
    const int uc = max(u, 0);
    const float* trow = loss.tm + (int64_t)uc * D + cb + 4 * kq;
#pragma unroll
    for (int t = 0; t < 4; ++t) tv[t] = *reinterpret_cast<const float4*>(trow + 16 * t);
    cf_raw = loss.coef[uc];
    cn_raw = loss.cnt_signed[uc];
    ls_cur = u;
  };
  auto fetch_rows = [&](int32_t row, int u) {
    const float4* src = reinterpret_cast<const float4*>(p + (int64_t)row * ld_p + 4 * kq);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = src[4 * i];
    const float4* dsrc = reinterpret_cast<const float4*>(dt + (int64_t)row * ld_dt + 4 * kq);
#pragma unroll
    for (int i = 0; i < 4; ++i) dv[i] = dsrc[4 * i];
    old_bits = *reinterpret_cast<const uint2*>(sign_io + (int64_t)slot_of(u) * 4 + 2 * half);
    if (RANK1) { ra_row = r1.ra[row]; rb_row = r1.rb[row]; }
  };
  if (u_lo < u_hi) {
    fetch_rows(row_n, u_lo);
    fetch_targets(ls_n);
    for (int u = u_lo; u < u_hi; ++u) {
      const int32_t row = row_n;
      const int s_a = min(u, n_units - 1) * 16 + r;
      const bool live = s_a < n_sel;
      const float livef = live ? 1.f : 0.f;
      row_n = row_nn;
      ls_n = ls_nn;
      row_nn = idx[slot_of(u + 2)];
      ls_nn = loss.slot[slot_of(u + 2)];
      // ---- P1: this wave's 64 columns of z;  P_dh: its 64 columns of the previous iteration's input gradient
      f32x4w acc[4], dacc[4];
      {
        float4 wq[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) wq[0][t] = wmine[(t * 8) * 64];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i + 1 < 8) {
#pragma unroll
            for (int t = 0; t < 4; ++t) wq[(i + 1) & 1][t] = wmine[(t * 8 + i + 1) * 64];
          }
          const float xv[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const float4 wf = wq[i & 1][t];
              const float wa = c == 0 ? wf.x : c == 1 ? wf.y : c == 2 ? wf.z : wf.w;
              if (i == 0 && c == 0) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, xv[0], f32x4w{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
              else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, xv[c], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) wq[0][t] = w2mine[(t * 4) * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i + 1 < 4) {
#pragma unroll
            for (int t = 0; t < 4; ++t) wq[(i + 1) & 1][t] = w2mine[(t * 4 + i + 1) * 64];
          }
          const float xv[4] = {dv[i].x, dv[i].y, dv[i].z, dv[i].w};
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const float4 wf = wq[i & 1][t];
              const float wa = c == 0 ? wf.x : c == 1 ? wf.y : c == 2 ? wf.z : wf.w;
              if (i == 0 && c == 0) dacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, xv[0], f32x4w{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
              else dacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, xv[c], dacc[t], 0, 0, 0);
            }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- the p rows into their transposition tile (rows past the end zeroed: no weight-gradient contribution)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        *reinterpret_cast<float4*>(tp + r * PTP + 16 * i + 4 * kq) = make_float4(livef * x[i].x, livef * x[i].y, livef * x[i].z, livef * x[i].w);
      // ---- loss gradient in the accumulator layout + the gated previous gradient; z out, sign bits, g into its tile
      const float cf = ls_cur >= 0 ? cf_raw : 0.f, cn = ls_cur >= 0 ? cn_raw : 0.f;
      float sq = 0.f;
      uint32_t bits[2] = {0u, 0u};
      const uint32_t ob[2] = {old_bits.x, old_bits.y};
      const float ra_c = ra_row, rb_c = rb_row;
      float* zrow = z + (int64_t)row * ld_z + cb + 4 * kq;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 zv = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        *reinterpret_cast<float4*>(zrow + 16 * t) = zv;
        bits[t >> 1] |= ((zv.x > 0.f ? 1u : 0u) | (zv.y > 0.f ? 2u : 0u) | (zv.z > 0.f ? 4u : 0u) | (zv.w > 0.f ? 8u : 0u)) << (16 * (t & 1) + 4 * kq);
        const float4 d4 = make_float4(zv.x - tv[t].x, zv.y - tv[t].y, zv.z - tv[t].z, zv.w - tv[t].w);
        sq = fmaf(d4.x, d4.x, sq); sq = fmaf(d4.y, d4.y, sq); sq = fmaf(d4.z, d4.z, sq); sq = fmaf(d4.w, d4.w, sq);
        const uint32_t m = ob[t >> 1] >> (16 * (t & 1) + 4 * kq);   // the previous pattern of these four columns
        float4 dq = make_float4(dacc[t][0], dacc[t][1], dacc[t][2], dacc[t][3]);
        if (RANK1) {
          dq.x = fmaf(rb_c, ub4[t].x, fmaf(ra_c, ua4[t].x, dq.x)); dq.y = fmaf(rb_c, ub4[t].y, fmaf(ra_c, ua4[t].y, dq.y));
          dq.z = fmaf(rb_c, ub4[t].z, fmaf(ra_c, ua4[t].z, dq.z)); dq.w = fmaf(rb_c, ub4[t].w, fmaf(ra_c, ua4[t].w, dq.w));
        }
        const float4 gv = make_float4(cf * d4.x + ((m & 1u) ? dq.x : 0.f), cf * d4.y + ((m & 2u) ? dq.y : 0.f),
                                      cf * d4.z + ((m & 4u) ? dq.z : 0.f), cf * d4.w + ((m & 8u) ? dq.w : 0.f));
        *reinterpret_cast<float4*>(tz + r * PTZ + 16 * t + 4 * kq) = gv;
      }
      sq *= livef;
      if (cn >= 0.f) ls0 = fmaf(cn, sq, ls0); else ls1 = fmaf(-cn, sq, ls1);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bits[q] |= (uint32_t)__shfl_xor((int)bits[q], 16);
        bits[q] |= (uint32_t)__shfl_xor((int)bits[q], 32);
      }
      if (live && kq == 0) *reinterpret_cast<uint2*>(sign_io + (int64_t)s_a * 4 + 2 * half) = make_uint2(bits[0], bits[1]);
      __builtin_amdgcn_sched_barrier(0);
      fetch_rows(row_n, u + 1);                                    // the next unit's operands into the registers just consumed
      fetch_targets(ls_n);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the tiles are wave-private: DS operations execute in issue order
      __builtin_amdgcn_wave_barrier();
      // ---- P3: this unit's weight-gradient product from the feature-major views (under the fetch just issued)
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        float a_op[8], b_op[4];
#pragma unroll
        for (int ta = 0; ta < 8; ++ta) a_op[ta] = tp[(4 * sp + kq) * PTP + 16 * ta + r];
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) b_op[tb] = tz[(4 * sp + kq) * PTZ + 16 * tb + r];
#pragma unroll
        for (int ta = 0; ta < 8; ++ta)
#pragma unroll
          for (int tb = 0; tb < 4; ++tb)
            gacc[ta * 4 + tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_op[ta], b_op[tb], gacc[ta * 4 + tb], 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the tiles are read before the next unit overwrites them)
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- block sum of the waves' D x H sums (pair order), loss sums
  __syncthreads();
  float* const red = wl;                                          // [4][D x H]
#pragma unroll
  for (int ta = 0; ta < 8; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int v = 0; v < 4; ++v) red[wave * D * H + (16 * ta + 4 * kq + v) * H + 16 * tb + r] = gacc[ta * 4 + tb][v];
  __syncthreads();
  for (int slot = blockIdx.x; slot < n_part; slot += gridDim.x) {
    float* const out = wg_partials + (int64_t)slot * D * D;
    const bool mine = slot == (int)blockIdx.x;
    for (int e = tid; e < D * D; e += 256) {
      const int i = e >> 7, c = e & 127, hf = c >> 6, ch = c & 63;
      out[e] = mine ? red[hf * D * H + i * H + ch] + red[(2 + hf) * D * H + i * H + ch] : 0.f;
    }
  }
  ls0 = wave_sum(ls0);
  ls1 = wave_sum(ls1);
  if (lane == 0) { lred[0][wave] = ls0; lred[1][wave] = ls1; }
  __syncthreads();
  if (tid == 0) {
    for (int slot = blockIdx.x; slot < n_part; slot += gridDim.x) {
      const bool mine = slot == (int)blockIdx.x;
      loss.partials[2 * slot + 0] = mine ? (lred[0][0] + lred[0][1]) + (lred[0][2] + lred[0][3]) : 0.f;
      loss.partials[2 * slot + 1] = mine ? (lred[1][0] + lred[1][1]) + (lred[1][2] + lred[1][3]) : 0.f;
    }
  }
}

static inline int del_fused_grid(int32_t n_sel) {
  const int n_tiles = (n_sel + 31) / 32;
  int grid = (n_tiles + 3) / 4;
  if (grid > 1024) grid = 1024;
  return grid < 1 ? 1 : grid;
}

}  // namespace gd

extern "C" int32_t gd_del_loss_bwd_blocks(int32_t n_sel) { return n_sel > 0 ? gd::del_fused_grid(n_sel) : 0; }

extern "C" int32_t gd_rows_gemm_wgrad_blocks(int32_t n_sel);

static int del_loss_bwd_impl(const float* p, int64_t ld_p, const int32_t* idx, int32_t n_sel, const float* w,
                             int32_t d, const int32_t* loss_slot, const float* tm, const float* coef,
                             const float* cnt_signed, float* dz, int64_t ld_dz, float* dp, int64_t ld_dp,
                             float* loss_partials, float* wgrad_partials, bool with_wgrad, void* stream, int32_t n_part_req = 0) {
  using namespace gd;
  const char* name = with_wgrad ? "gd_del_loss_bwd_wgrad_f32" : "gd_del_loss_bwd_f32";
  if (n_sel == 0) return GD_OK;
  GD_REQUIRE(p && idx && w && loss_slot && tm && coef && cnt_signed && dp && loss_partials && (with_wgrad ? wgrad_partials != nullptr : dz != nullptr),
             GD_E_NULL, "%s: null pointer", name);
  GD_REQUIRE(d == 32 || d == 64, GD_E_DIM, "%s: d=%d must be 32 or 64", name, d);
  GD_REQUIRE(ld_p >= d && (!dz || ld_dz >= d) && ld_dp >= d && ld_p % 4 == 0 && (!dz || ld_dz % 4 == 0) && ld_dp % 4 == 0, GD_E_DIM,
             "%s: bad row strides", name);
  GD_REQUIRE(aligned16(p) && aligned16(tm) && (!dz || aligned16(dz)) && aligned16(dp) && (!with_wgrad || aligned16(wgrad_partials)), GD_E_ALIGN,
             "%s: unaligned", name);
  GD_REQUIRE(p != dp && p != dz && dz != dp, GD_E_DIM, "%s: buffers must not alias", name);
  hipStream_t s = (hipStream_t)stream;
  const DelLoss loss{loss_slot, tm, coef, cnt_signed, loss_partials};
  if (!with_wgrad) {
    const dim3 grid(del_fused_grid(n_sel));
    const size_t lds = ((size_t)d * d + (size_t)d * (d + 4)) * sizeof(float);     // the two weight images
    switch (d / 32) {
      case 1: hipLaunchKernelGGL((del_loss_bwd_kernel<1, false>), grid, dim3(256), lds, s, p, ld_p, idx, n_sel, w, loss, dz, ld_dz, dp, ld_dp, nullptr); break;
      default: hipLaunchKernelGGL((del_loss_bwd_kernel<2, false>), grid, dim3(256), lds, s, p, ld_p, idx, n_sel, w, loss, dz, ld_dz, dp, ld_dp, nullptr); break;
    }
    return launched("del_loss_bwd");
  }
  // one block per partial matrix of the weight-gradient reduction; + the four waves' transposition tiles (the block sum of
  // the D x D accumulators reuses the whole allocation: 4 D^2 floats <= what is there)
  static const bool ws_on = [] { const char* e = getenv("GD_DEL2_WS"); return !(e && atoi(e) == 0); }();
  if (ws_on && d == 64 && n_sel >= 65536) {       // weight-stationary form: one wave per SIMD, 16-row units (above)
    // n_part_req (gd_del_loss_bwd_wgrad_parts_f32): fill that many partial slots (>= this launch's blocks) instead of
    // gd_rows_gemm_wgrad_blocks(n_sel) of them
    const int n_all = gd_rows_gemm_wgrad_blocks(n_sel);
    const int n_part = n_part_req > 0 ? n_part_req : n_all;
    GD_REQUIRE(n_part <= n_all && n_part >= (ws_cu_count() < n_all ? ws_cu_count() : n_all), GD_E_DIM,
               "%s: n_part=%d outside [gd_del_loss_bwd_wgrad_parts, gd_rows_gemm_wgrad_blocks = %d]", name, n_part, n_all);
    const int grid_ws = ws_cu_count() < n_part ? ws_cu_count() : n_part;
    constexpr int kLdsWs = 4 * 64 * 64 * 4;
    if (dz) {
      static const hipError_t a1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&del_loss_bwd_ws_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsWs);
      if (a1 != hipSuccess) return fail(-(int)a1, "gd_del_loss_bwd_wgrad_f32: %s", hipGetErrorString(a1));
      hipLaunchKernelGGL((del_loss_bwd_ws_kernel<true>), dim3(grid_ws), dim3(256), kLdsWs, s, p, ld_p, idx, n_sel, w, loss, dz, ld_dz, dp, ld_dp,
                         wgrad_partials, n_part);
    } else {
      static const hipError_t a0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&del_loss_bwd_ws_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsWs);
      if (a0 != hipSuccess) return fail(-(int)a0, "gd_del_loss_bwd_wgrad_f32: %s", hipGetErrorString(a0));
      hipLaunchKernelGGL((del_loss_bwd_ws_kernel<false>), dim3(grid_ws), dim3(256), kLdsWs, s, p, ld_p, idx, n_sel, w, loss, dz, ld_dz, dp, ld_dp,
                         wgrad_partials, n_part);
    }
    return launched("del_loss_bwd_ws");
  }
  GD_REQUIRE(n_part_req == 0 || n_part_req == gd_rows_gemm_wgrad_blocks(n_sel), GD_E_DIM,
             "%s: n_part=%d but this size leaves gd_rows_gemm_wgrad_blocks(n_sel) = %d partials", name, n_part_req, gd_rows_gemm_wgrad_blocks(n_sel));
  const dim3 grid(gd_rows_gemm_wgrad_blocks(n_sel));
#define GD_DF_CASE(NT_)                                                                                                        \
  do {                                                                                                                         \
    constexpr int kD = 32 * NT_;                                                                                               \
    constexpr int kLds = (kD * kD + kD * (kD + 4) + 4 * 32 * (kD + 4)) * 4;                                                    \
    static_assert(kLds >= 4 * kD * kD * 4, "block sum needs 4 D^2 floats");                                                    \
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&del_loss_bwd_kernel<NT_, true>),         \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds);                      \
    if (attr != hipSuccess) return fail(-(int)attr, "gd_del_loss_bwd_wgrad_f32: %s", hipGetErrorString(attr));                 \
    hipLaunchKernelGGL((del_loss_bwd_kernel<NT_, true>), grid, dim3(256), kLds, s, p, ld_p, idx, n_sel, w, loss, dz, ld_dz,    \
                       dp, ld_dp, wgrad_partials);                                                                             \
  } while (0)
  if (d == 32) GD_DF_CASE(1); else GD_DF_CASE(2);
#undef GD_DF_CASE
  return launched("del_loss_bwd_wgrad");
}

extern "C" int gd_del_loss_bwd_f32(const float* p, int64_t ld_p, const int32_t* idx, int32_t n_sel, const float* w,
                                   int32_t d, const int32_t* loss_slot, const float* tm, const float* coef,
                                   const float* cnt_signed, float* dz, int64_t ld_dz, float* dp, int64_t ld_dp,
                                   float* loss_partials, void* stream) {
  return del_loss_bwd_impl(p, ld_p, idx, n_sel, w, d, loss_slot, tm, coef, cnt_signed, dz, ld_dz, dp, ld_dp, loss_partials, nullptr,
                           false, stream);
}

extern "C" int gd_del_loss_bwd_wgrad_f32(const float* p, int64_t ld_p, const int32_t* idx, int32_t n_sel, const float* w,
                                         int32_t d, const int32_t* loss_slot, const float* tm, const float* coef,
                                         const float* cnt_signed, float* dz, int64_t ld_dz, float* dp, int64_t ld_dp,
                                         float* loss_partials, float* wgrad_partials, void* stream) {
  return del_loss_bwd_impl(p, ld_p, idx, n_sel, w, d, loss_slot, tm, coef, cnt_signed, dz, ld_dz, dp, ld_dp, loss_partials,
                           wgrad_partials, true, stream);
}

// partial matrices gd_del_loss_bwd_wgrad_parts_f32 has blocks for (the weight-stationary form: one per compute unit)
extern "C" int32_t gd_del_loss_bwd_wgrad_parts(int32_t n_sel, int32_t d) {
  if (n_sel <= 0) return 0;
  const int nb = gd_rows_gemm_wgrad_blocks(n_sel);
  static const bool ws_on = [] { const char* e = getenv("GD_DEL2_WS"); return !(e && atoi(e) == 0); }();
  if (ws_on && d == 64 && n_sel >= 65536) return gd::ws_cu_count() < nb ? gd::ws_cu_count() : nb;
  return nb;
}

extern "C" int gd_del_loss_bwd_wgrad_parts_f32(const float* p, int64_t ld_p, const int32_t* idx, int32_t n_sel, const float* w, int32_t d,
                                               const int32_t* loss_slot, const float* tm, const float* coef, const float* cnt_signed,
                                               float* dz, int64_t ld_dz, float* dp, int64_t ld_dp, float* loss_partials,
                                               float* wgrad_partials, int32_t n_part, void* stream) {
  GD_REQUIRE(n_part > 0 || n_sel == 0, GD_E_DIM, "gd_del_loss_bwd_wgrad_parts_f32: n_part=%d", n_part);
  return del_loss_bwd_impl(p, ld_p, idx, n_sel, w, d, loss_slot, tm, coef, cnt_signed, dz, ld_dz, dp, ld_dp, loss_partials,
                           wgrad_partials, true, stream, n_part);
}

extern "C" int32_t gd_del1_loss_wgrad_covers(int32_t n_sel, int32_t d) {
  static const bool on = [] { const char* e = getenv("GD_DEL1_FUSED"); return !(e && atoi(e) == 0); }();
  return on && d == 128 && n_sel >= 65536 ? 1 : 0;
}

extern "C" int32_t gd_del1_loss_wgrad_parts(int32_t n_sel) {
  if (n_sel <= 0) return 0;
  const int nb = gd_rows_gemm_wgrad_blocks(n_sel);
  return gd::ws_cu_count() < nb ? gd::ws_cu_count() : nb;
}

extern "C" int gd_del1_loss_wgrad_f32(const float* p, int64_t ld_p, const int32_t* idx, int32_t n_sel, const float* w, int32_t d,
                                      float* z, int64_t ld_z, uint32_t* sign_out, const int32_t* loss_slot, const float* tm,
                                      const float* coef, const float* cnt_signed, const float* g_add, int64_t ld_gadd,
                                      float* loss_partials, float* wgrad_partials, int32_t n_part, void* stream) {
  using namespace gd;
  const char* name = "gd_del1_loss_wgrad_f32";
  if (n_sel == 0) return GD_OK;
  GD_REQUIRE(p && idx && w && z && sign_out && loss_slot && tm && coef && cnt_signed && loss_partials && wgrad_partials, GD_E_NULL,
             "%s: null pointer", name);
  GD_REQUIRE(d == 128, GD_E_DIM, "%s: d=%d must be 128 (other widths: gd_rows_gemm_signs_f32 + gd_rows_gemm_wgrad_loss_f32)", name, d);
  GD_REQUIRE(ld_p >= d && ld_z >= d && ld_p % 4 == 0 && ld_z % 4 == 0 && (!g_add || (ld_gadd >= d && ld_gadd % 4 == 0)), GD_E_DIM,
             "%s: bad row strides", name);
  GD_REQUIRE(aligned16(p) && aligned16(z) && aligned16(tm) && aligned16(w) && (!g_add || aligned16(g_add)) && aligned16(wgrad_partials),
             GD_E_ALIGN, "%s: unaligned", name);
  GD_REQUIRE(p != z && g_add != z, GD_E_DIM, "%s: z must not alias p or g_add (the two waves of a pair read whole rows)", name);
  hipStream_t s = (hipStream_t)stream;
  const DelLoss loss{loss_slot, tm, coef, cnt_signed, loss_partials};
  GD_REQUIRE(n_part >= 1 && n_part <= gd_rows_gemm_wgrad_blocks(n_sel), GD_E_DIM,
             "%s: n_part=%d outside [1, gd_rows_gemm_wgrad_blocks(n_sel)=%d]", name, n_part, gd_rows_gemm_wgrad_blocks(n_sel));
  const int grid = ws_cu_count() < n_part ? ws_cu_count() : n_part;
  constexpr int kLds = 4 * 128 * 64 * 4;
  if (g_add) {
    static const hipError_t a1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&del1_loss_wgrad_ws_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (a1 != hipSuccess) return fail(-(int)a1, "%s: %s", name, hipGetErrorString(a1));
    hipLaunchKernelGGL((del1_loss_wgrad_ws_kernel<true>), dim3(grid), dim3(256), kLds, s, p, ld_p, idx, n_sel, w, z, ld_z, sign_out, loss, g_add,
                       ld_gadd, wgrad_partials, n_part);
  } else {
    static const hipError_t a0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&del1_loss_wgrad_ws_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (a0 != hipSuccess) return fail(-(int)a0, "%s: %s", name, hipGetErrorString(a0));
    hipLaunchKernelGGL((del1_loss_wgrad_ws_kernel<false>), dim3(grid), dim3(256), kLds, s, p, ld_p, idx, n_sel, w, z, ld_z, sign_out, loss, nullptr,
                       0, wgrad_partials, n_part);
  }
  return launched("del1_loss_wgrad_ws");
}

extern "C" int gd_del1_chain_loss_wgrad_f32(const float* p, int64_t ld_p, const int32_t* idx, int32_t n_sel, const float* w, int32_t d,
                                            float* z, int64_t ld_z, uint32_t* sign_io, const int32_t* loss_slot, const float* tm,
                                            const float* coef, const float* cnt_signed, const float* dt, int64_t ld_dt, int32_t d_next,
                                            const float* w_next, const float* row_a, const float* col_a, const float* row_b,
                                            const float* col_b, float* loss_partials, float* wgrad_partials, int32_t n_part,
                                            void* stream) {
  using namespace gd;
  const char* name = "gd_del1_chain_loss_wgrad_f32";
  if (n_sel == 0) return GD_OK;
  GD_REQUIRE(p && idx && w && z && sign_io && loss_slot && tm && coef && cnt_signed && dt && w_next && loss_partials && wgrad_partials, GD_E_NULL,
             "%s: null pointer", name);
  GD_REQUIRE(d == 128 && d_next == 64, GD_E_DIM, "%s: widths %d / %d must be 128 / 64", name, d, d_next);
  GD_REQUIRE(ld_p >= d && ld_z >= d && ld_dt >= d_next && ld_p % 4 == 0 && ld_z % 4 == 0 && ld_dt % 4 == 0, GD_E_DIM, "%s: bad row strides", name);
  GD_REQUIRE(aligned16(p) && aligned16(z) && aligned16(tm) && aligned16(dt) && aligned16(wgrad_partials) && (reinterpret_cast<uintptr_t>(sign_io) & 7u) == 0,
             GD_E_ALIGN, "%s: unaligned", name);
  GD_REQUIRE(p != z && dt != z, GD_E_DIM, "%s: z must not alias p or dt", name);
  GD_REQUIRE(n_part >= 1 && n_part <= gd_rows_gemm_wgrad_blocks(n_sel), GD_E_DIM,
             "%s: n_part=%d outside [1, gd_rows_gemm_wgrad_blocks(n_sel)=%d]", name, n_part, gd_rows_gemm_wgrad_blocks(n_sel));
  const DelLoss loss{loss_slot, tm, coef, cnt_signed, loss_partials};
  const int grid = ws_cu_count() < n_part ? ws_cu_count() : n_part;
  constexpr int kLds = (4 * (16 * 144 + 16 * 80) + 128 * 128 + 64 * 128) * 4;      // tiles + W_D image + W_next image = 155,648 B (>= the 128 KB block sums)
  const bool rank1 = row_a || col_a || row_b || col_b;
  GD_REQUIRE(!rank1 || (row_a && col_a && row_b && col_b && aligned16(col_a) && aligned16(col_b)), GD_E_NULL,
             "%s: the rank-1 terms come as four pointers (row_a, col_a, row_b, col_b; 16-byte aligned column vectors)", name);
  const ChainRank1 r1{row_a, col_a, row_b, col_b};
  if (rank1) {
    static const hipError_t at = hipFuncSetAttribute(reinterpret_cast<const void*>(&del1_chain_ws_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (at != hipSuccess) return fail(-(int)at, "%s: %s", name, hipGetErrorString(at));
    hipLaunchKernelGGL(del1_chain_ws_kernel<true>, dim3(grid), dim3(256), kLds, (hipStream_t)stream, p, ld_p, idx, n_sel, w, z, ld_z, sign_io, loss, dt,
                       ld_dt, w_next, wgrad_partials, n_part, r1);
  } else {
    static const hipError_t at = hipFuncSetAttribute(reinterpret_cast<const void*>(&del1_chain_ws_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (at != hipSuccess) return fail(-(int)at, "%s: %s", name, hipGetErrorString(at));
    hipLaunchKernelGGL(del1_chain_ws_kernel<false>, dim3(grid), dim3(256), kLds, (hipStream_t)stream, p, ld_p, idx, n_sel, w, z, ld_z, sign_io, loss, dt,
                       ld_dt, w_next, wgrad_partials, n_part, r1);
  }
  return launched("del1_chain_ws");
}
