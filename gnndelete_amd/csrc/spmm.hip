// CSR SpMM / gather-scatter-add kernels for the frozen-backbone message passing
// (GCN weighted sum, GIN plain sum, R-GCN per-relation mean).
//
// Layout: CSR over TARGET rows, col[] = source node, fp32 row-major features.
// Mapping to CDNA4: one 64-lane wave owns one output row.  A feature row of d floats is d/4
// float4 vectors; LPR lanes (16 B each -> one fully coalesced 16*LPR-byte read per neighbour)
// cover it, so G = 64/LPR neighbours are gathered concurrently by the G lane groups of the
// wave.  The (col, val) pairs of a row are fetched 64 at a time with one coalesced load and
// handed to the groups with ds_bpermute (__shfl) - the LDS crossbar, no LDS storage.  Group
// partial sums are combined with a log2(G)-step xor-shuffle tree, so the summation order is
// fixed (bit-reproducible run to run).  HBM-bound: bytes = 4(n+1) + 8 nnz + 8 n d.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace gd {

template <int LPR, int VPL, bool MEAN>
__global__ __launch_bounds__(256) void spmm_vec_kernel(const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col,
                                                       const float* __restrict__ val,
                                                       const float* __restrict__ x, int64_t ldx,
                                                       float* __restrict__ y, int64_t ldy,
                                                       const float* __restrict__ bias, float self_coef,
                                                       int32_t n_rows, int32_t x_rows_mod, int32_t d4) {
  constexpr int G = kWave / LPR;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int g = lane / LPR;
  const int li = lane % LPR;
  const int start = rowptr[row];
  const int end = rowptr[row + 1];

  float4 acc[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = f4_zero();

  for (int base = start; base < end; base += kWave) {
    const int k = base + lane;
    const bool live = k < end;
    const int c = live ? col[k] : 0;
    const float w = live ? (val ? val[k] : 1.0f) : 0.0f;
    const int cnt = min(kWave, end - base);
    // padded slots carry w = 0 and c = 0 (a harmless re-read of row 0), so the trip count can
    // be rounded up to the unroll factor
    const int trips = (cnt + G - 1) / G;

    for (int it = 0; it < trips; ++it) {
      const int j = it * G + g;
      const int cj = __shfl(c, j);
      const float wj = __shfl(w, j);
      const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)cj * ldx);
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int vec = li + v * LPR;
        if (vec < d4) acc[v] = f4_fma(wj, xr[vec], acc[v]);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) acc[v] = f4_add(acc[v], f4_shfl_xor(acc[v], off));
  }
  if (g != 0) return;
  const float scale = MEAN ? (end > start ? 1.0f / (float)(end - start) : 0.0f) : 1.0f;
  // MEAN mode (typed SpMM): virtual row = rel * n + i, self/bias unused
  const int self_row = x_rows_mod > 0 ? row % x_rows_mod : row;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int vec = li + v * LPR;
    if (vec >= d4) continue;
    float4 o = acc[v];
    if (MEAN) {
      o.x *= scale; o.y *= scale; o.z *= scale; o.w *= scale;
    } else {
      if (self_coef != 0.0f) {
        const float4 s = reinterpret_cast<const float4*>(x + (int64_t)self_row * ldx)[vec];
        o = f4_fma(self_coef, s, o);
      }
      if (bias) o = f4_add(o, reinterpret_cast<const float4*>(bias)[vec]);
    }
    reinterpret_cast<float4*>(y + (int64_t)row * ldy)[vec] = o;
  }
}

// ---------------------------------------------------------------------------------------------
// Load-balanced variant: the CSR is cut into work items of at most 64 in-edges (one coalesced
// (col,val) fetch each).  A heavy-tailed graph otherwise serialises its hub rows on single
// waves (a degree-2000 row = 1000 dependent gather rounds = the whole kernel's run time).
// Items that cover a whole row write y directly; the pieces of a split row go to `scratch` and
// a second tiny kernel adds them up in slot order - no atomics, bit-reproducible.
// items[i] = {row, start, end, slot (-1 = whole row)};  split[i] = {row, first_slot, n_slots, 0}.
// ---------------------------------------------------------------------------------------------
// Persistent, XCD-windowed variant.  The grid is exactly the resident set (8 XCDs x 32 CUs x 8
// blocks); XCD k owns the k-th contiguous eighth of the work items and its 1024 resident waves
// sweep that range TOGETHER: wave w visits items base + t*1024 + w, so at any moment one XCD is
// working inside a window of ~1024 consecutive rows.  With a locality-preserving node order the
// feature rows those rows gather are shared and stay in that XCD's private 4 MB L2, instead of
// every edge going out to the fabric (measured: 18 % L2 hit rate and 5.4x the algorithmic bytes
// with one block per 64 random rows).
// ADDR32: byte offsets of x rows fit 32 bits and (row id, row pitch in bytes) fit 24 bits, so a row
// address is ONE full-rate v_mul_u32_u24 instead of a 64-bit multiply (three quarter-rate integer
// multiplies per gathered row made the kernel issue-bound, not memory-bound, once the rows hit in L2).
// Latency structure (what the kernel is actually bound by - a wave spends its life waiting on memory, so the
// number of DEPENDENT round trips per visited row is the cost): the item descriptor is fetched two visits
// ahead and the item's (col, val) one visit ahead and the bias is requested together with the gathers (as the
// accumulators' initial value), so a GCN row costs ONE round trip (its gathers) instead of three (descriptor ->
// indices, gathers, then the bias behind them); only a self / residual row (GIN, GraphSAGE) is still fetched
// in the epilogue - holding it across the gathers costs the registers that decide 8 vs 7 waves per SIMD.
// xcd_bounds (optional, 9 ints): XCD k sweeps the items [xcd_bounds[k], xcd_bounds[k+1]) instead of an equal eighth -
// the host balances the ranges by bytes moved (a self-loop-only row costs two row transfers, a 64-edge piece 65),
// so that no XCD's fabric link idles while another still has a third of its traffic to go.
// AUX (GATConv's backward, gat.hip): the edge values are not read from `val` but THROUGH a permutation from an interleaved
// (weight, addend) array in another edge order - w[k] = aux[perm[k]].x - and the addends of a row's entries are summed into
// aux_sum[row]: the transposition pass that used to make both (gat_transpose_edge_kernel, 33 us of the 775 us GAT step: 1.84 M
// random 8-byte gathers at 8 lanes per row) disappears into this kernel, where the same gather is one more load per visit
// that arrives behind the row gathers.  perm is fetched a visit ahead with the indices, aux at the top of the visit.
template <int LPR, int VPL, int U, bool EXACT, bool ADDR32, bool AUX = false, bool NOVAL = false>
__device__ __forceinline__ void spmm_persist_body(const int4* __restrict__ items, int32_t n_items,
                                                  const int32_t* __restrict__ xcd_bounds,
                                                  const int32_t* __restrict__ col, const float* __restrict__ val,
                                                  const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                  int64_t ldy, const float* __restrict__ bias, float self_coef,
                                                  const float* __restrict__ xs, float* __restrict__ scratch, int32_t d4,
                                                  int32_t nnz, const int32_t* __restrict__ perm = nullptr,
                                                  const float2* __restrict__ aux = nullptr, float* __restrict__ aux_sum = nullptr) {
  constexpr int G = kWave / LPR;
  constexpr int kXcd = 8;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR, li = lane % LPR;
  const int xcd = blockIdx.x % kXcd;
  const int stride = (gridDim.x / kXcd) * 4;                        // waves per XCD
  const int wx = (blockIdx.x / kXcd) * 4 + (threadIdx.x >> 6);
  const int per = (n_items + kXcd - 1) / kXcd;
  const int i0 = xcd_bounds ? xcd_bounds[xcd] : xcd * per;
  const int i1 = xcd_bounds ? xcd_bounds[xcd + 1] : min(n_items, i0 + per);
  // One-launch form (no scratch rows: hub rows are GROUP items whose four member waves meet at block barriers further down): that
  // is only block-uniform when every range limit is a multiple of 4 - a table that breaks the invariant aborts the launch here
  // instead of deadlocking it (include/gnndelete_hip.h states the invariant; in the prologue, where it costs the sweep
  // no register - inside the group branch it took the 64- and 128-float kernels from 7 / 8 to 6 / 7 waves per SIMD).  The piece
  // form (gd_spmm_csr_balanced_f32, always a non-NULL scratch pointer) has no group items and takes limits cut anywhere.
  if (xcd_bounds && !scratch && ((i0 | i1) & 3)) __builtin_trap();

  // per-lane constants: byte offset of this lane's vectors inside a row, its slice of the bias
  uint32_t lo[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) lo[v] = 16u * (uint32_t)(EXACT ? li + v * LPR : min(li + v * LPR, d4 - 1));
  const char* bb = reinterpret_cast<const char*>(bias);
  const uint32_t pitch_b = (uint32_t)ldx * 4u;
  const char* xb = reinterpret_cast<const char*>(x);

  int i = i0 + wx;
  if (i >= i1) return;

  // Descriptors are wave-uniform: ONE lane fetches them (a 64-lane load of one address would still occupy the
  // texture addresser like a full gather) and they are kept in SGPRs, so the trip structure and the
  // epilogue's addressing are scalar work.
  struct Desc { int row, start, end, slot; };
  auto uniform = [](const int4& v) {
    Desc d;
    d.row = __builtin_amdgcn_readfirstlane(v.x);
    d.start = __builtin_amdgcn_readfirstlane(v.y);
    d.end = __builtin_amdgcn_readfirstlane(v.z);
    d.slot = __builtin_amdgcn_readfirstlane(v.w);
    return d;
  };
  int4 dv = make_int4(0, 0, 0, 0), dv1 = dv;
  if (lane == 0) {
    dv = items[i];
    dv1 = items[min(i + stride, i1 - 1)];
  }
  Desc d0 = uniform(dv), d1 = uniform(dv1);
  int kk = min(d0.start + lane, nnz - 1);
  int c = col[kk];
  float w = (AUX || NOVAL) ? 0.f : (val ? val[kk] : 1.0f);
  int pk = AUX ? perm[kk] : 0;
  // One-launch form (gd_spmm_csr_onepass_f32): slot == -2 marks a GROUP member - the four waves of a block visit four
  // consecutive, 4-aligned items at the same time, and a hub row (more than 64 in-edges) is laid out as such a
  // quadruple, member w = wave w's own contiguous share [start, end) of the row's edges (walked 64 at a time: the
  // first chunk rides the sweep's prefetch pipeline like any item, further chunks - rows above 256 in-edges - are
  // fetched in place).  The four partial rows meet in LDS and wave 0 adds them in wave order (+ self term + bias) and
  // writes the row: no scratch rows, no fix-up launch, no atomics, a fixed summation order.  row < 0 = padding.
  __shared__ float4 grp_red[4][LPR * VPL];
  __shared__ float grp_aux[4];
  for (; i < i1; i += stride) {
    const int row = d0.row, slot = d0.slot;
    int base = d0.start;
    int c_cur = c;
    float w_raw = w;
    float a_raw = 0.f;                                   // AUX: this lane's addend
    if (AUX) {
      const float2 ad = aux[pk];
      w_raw = ad.x;
      a_raw = ad.y;
    }
    // prefetch (clamped, branch-free): descriptor of the visit after next, indices of the next visit
    if (lane == 0) dv = items[min(i + 2 * stride, i1 - 1)];
    kk = min(d1.start + lane, nnz - 1);
    c = col[kk];
    if (AUX) pk = perm[kk];
    else if (!NOVAL) w = val ? val[kk] : 1.0f;
    float a_sum = 0.f;                                   // AUX: sum of the addends of a group member's chunks
    float4 acc[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = f4_zero();
    // MULTI-ROW item (slot <= -16; MAXR rows at most, by width): up to four CONSECUTIVE light rows (row .. row + nr - 1, adjacent
    // in the CSR, <= 64 in-edges together) share this visit.  A visit is one dependent round trip however few rows it gathers
    // and the sweep is latency-bound (7.7 in-edges per row on the bench graph, while a wave can keep U x G = 16 gathers of a
    // 64-float row in flight), so packing rows multiplies the gathers in flight and divides the visits.  Every extra row costs
    // an accumulator: 4 rows per item take this kernel from 64 to 88 VGPRs (8 -> 5 waves per SIMD), which costs the
    // single-row items more than the packing gains (measured: 668 us per step either way); so the 128-float kernel, whose
    // U x G = 8 is already reached by one average row, keeps one row per item at 8 waves, the 64-float kernel takes PAIRS
    // (15 edges ~ its 16 gathers) for one more accumulator, narrower rows take four.
    // v = -slot - 16 = cumulative edge counts after row 0 / 1 / 2 (7 bits each) | (nr - 1) << 21; slot j of the combined edge
    // list belongs to row (j >= b1) + (j >= b2) + (j >= b3).  A row's edges are dealt to the lane groups by their position in
    // the ITEM, so its sum is associated differently from the single-row form's - deterministic, equal to fp32 rounding.
    constexpr int kFar = 1 << 20;
    constexpr int W4 = LPR * VPL;                       // float4 vectors of a row this instantiation covers
    constexpr int MAXR = W4 >= 32 ? 1 : (W4 == 16 ? 2 : 4);
    const bool multi = MAXR > 1 && slot <= -16;
    int nr = 1, b1 = kFar, b2 = kFar, b3 = kFar;
    if (multi) {
      const int v = -slot - 16;
      nr = ((v >> 21) & 3) + 1;
      b1 = v & 127;
      b2 = (MAXR > 2 && nr > 2) ? (v >> 7) & 127 : kFar;
      b3 = (MAXR > 3 && nr > 3) ? (v >> 14) & 127 : kFar;
    }
    float4 mb[MAXR > 1 ? MAXR - 1 : 1][VPL];           // accumulators of rows 1 .. MAXR - 1 of a multi-row item (row 0: acc)
#pragma unroll
    for (int q = 0; q < (MAXR > 1 ? MAXR - 1 : 1); ++q)
#pragma unroll
      for (int v = 0; v < VPL; ++v) mb[q][v] = f4_zero();
    auto add_multi = [&](int j, float wv, const float4(&xr)[VPL]) {
      const int r = (j >= b1) + (MAXR > 2 ? (j >= b2) + (j >= b3) : 0);
      const float w0 = r == 0 ? wv : 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) acc[v] = f4_fma(w0, xr[v], acc[v]);
#pragma unroll
      for (int q = 0; q < MAXR - 1; ++q) {
        const float wq = r == q + 1 ? wv : 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) mb[q][v] = f4_fma(wq, xr[v], mb[q][v]);
      }
    };
    const bool whole = (slot == -1 || multi) && row >= 0, self = whole && !multi && self_coef != 0.0f;
    // the accumulators of lane group 0 start from the bias (requested with the gathers: no round trip of its
    // own, no registers held across rows); the pieces of a split row get theirs in the fix-up / group sum
    if (bias && whole && !multi && g == 0) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) acc[v] = *reinterpret_cast<const float4*>(bb + lo[v]);
    }
    // likewise a self / residual row with coefficient 1 (GraphSAGE's root term, GIN with eps = 0) is the
    // initial value of lane group 1's accumulators; any other coefficient (or a row as wide as the wave, where
    // there is no second group) is applied in the epilogue, one more round trip
    const bool self_in_acc = self && G >= 2 && self_coef == 1.0f;
    if (self_in_acc && g == 1) {
      const char* sb0 = reinterpret_cast<const char*>(xs + (int64_t)row * ldx);
#pragma unroll
      for (int v = 0; v < VPL; ++v) acc[v] = *reinterpret_cast<const float4*>(sb0 + lo[v]);
    }
    for (;;) {       // 64-edge chunks of the item: one, except for the members of a hub row above 256 in-edges
      const int cnt = min(kWave, d0.end - base);
      const float w_cur = lane < cnt ? w_raw : 0.f;
      const int trips = (cnt + G - 1) / G;
      // Slot j of the item (j = trip * G + g) is fetched from lane j of (c_cur, w_cur) through the LDS
      // crossbar (byte-addressed ds_bpermute, everything kept pre-multiplied by 4).  A wave64 x 16-byte load
      // occupies the CU's texture addresser for 16 cycles whether or not its rows are useful, so trips past
      // the end of the item are never issued (scalar trip control); only the last, partly filled trip pads:
      // its extra lane groups re-read the item's last neighbour (cached) and take weight 0 from lane `cnt`.
      const int last4 = 4 * cnt - 4, end4 = 4 * cnt;
      auto fetch = [&](int t, float4(&xv)[VPL], float& wj) {
        const int j4 = 4 * (t * G) + 4 * g;
        const int cs = __builtin_amdgcn_ds_bpermute(min(j4, last4), c_cur);
        // (NOVAL: an unweighted sum - the row scales of a normalised adjacency live in the producers' / this kernel's epilogue -
        //  has no weight stream: a slot is live or padding, decided from its position)
        if (NOVAL) wj = j4 < end4 ? 1.0f : 0.f;
        else wj = __int_as_float(__builtin_amdgcn_ds_bpermute(min(j4, end4), __float_as_int(w_cur)));
        if (ADDR32) {   // row byte offset by one full-rate 24-bit multiply, 32-bit offset on a scalar base
          const uint32_t ro = __umul24((uint32_t)cs, pitch_b);
  #pragma unroll
          for (int v = 0; v < VPL; ++v) xv[v] = *reinterpret_cast<const float4*>(xb + (ro + lo[v]));
        } else {
          const char* xr = reinterpret_cast<const char*>(x + (int64_t)cs * ldx);
  #pragma unroll
          for (int v = 0; v < VPL; ++v) xv[v] = *reinterpret_cast<const float4*>(xr + lo[v]);
        }
      };
      int t0 = 0;
      for (; t0 + U <= trips; t0 += U) {
        float4 xv[U][VPL];
        float wj[U];
  #pragma unroll
        for (int u = 0; u < U; ++u) fetch(t0 + u, xv[u], wj[u]);
        if (!multi) {                              // (wave-uniform)
  #pragma unroll
          for (int u = 0; u < U; ++u)
  #pragma unroll
            for (int v = 0; v < VPL; ++v) acc[v] = f4_fma(wj[u], xv[u][v], acc[v]);
        } else {
  #pragma unroll
          for (int u = 0; u < U; ++u) add_multi((t0 + u) * G + g, wj[u], xv[u]);
        }
      }
      if (U > 1) {
        const int rem = trips - t0;
        if (rem > 0) {
          float4 xv[U - 1 > 0 ? U - 1 : 1][VPL];
          float wj[U - 1 > 0 ? U - 1 : 1];
  #pragma unroll
          for (int u = 0; u < U - 1; ++u)
            if (u < rem) fetch(t0 + u, xv[u], wj[u]);
  #pragma unroll
          for (int u = 0; u < U - 1; ++u)
            if (u < rem) {
              if (!multi) {
  #pragma unroll
                for (int v = 0; v < VPL; ++v) acc[v] = f4_fma(wj[u], xv[u][v], acc[v]);
              } else {
                add_multi((t0 + u) * G + g, wj[u], xv[u]);
              }
            }
        }
      }
      if (AUX && slot == -2) a_sum += wave_sum(lane < cnt ? a_raw : 0.f);
      base += kWave;
      if (base >= d0.end) break;
      const int k2 = min(base + lane, nnz - 1);
      c_cur = col[k2];
      if (AUX) {
        const float2 ad = aux[perm[k2]];
        w_raw = ad.x;
        a_raw = ad.y;
      } else if (!NOVAL) {
        w_raw = val ? val[k2] : 1.0f;
      }
    }
    if (multi) {
      if (AUX) {                                   // (two rows at most where AUX is built: 64 floats and wider)
        const int cnt0 = d0.end - d0.start;
        const float s0 = wave_sum(lane < min(b1, cnt0) ? a_raw : 0.f), s1 = wave_sum((lane >= b1 && lane < cnt0) ? a_raw : 0.f);
        if (lane == 0) {
          aux_sum[row] = s0;
          if (nr > 1) aux_sum[row + 1] = s1;
        }
      }
#pragma unroll
      for (int q = 0; q < MAXR; ++q) {
        if (q < nr) {                              // (wave-uniform)
#pragma unroll
          for (int v = 0; v < VPL; ++v) {
            float4 o = f4_group_sum<LPR>(q == 0 ? acc[v] : mb[(q > 0 && q < MAXR) ? q - 1 : 0][v]);
            if (g == 0 && (EXACT || li + v * LPR < d4)) {
              const int64_t rq = (int64_t)row + q;
              if (self_coef != 0.0f)
                o = f4_fma(self_coef, *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(xs + rq * ldx) + lo[v]), o);
              if (bias) o = f4_add(o, *reinterpret_cast<const float4*>(bb + lo[v]));
              *reinterpret_cast<float4*>(reinterpret_cast<char*>(y + rq * ldy) + lo[v]) = o;
            }
          }
        }
      }
      d0 = d1;
      d1 = uniform(dv);
      continue;
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = f4_group_sum<LPR>(acc[v]);
    if (AUX && slot == -1 && row >= 0) {          // a whole row in one item: its addends' sum
      const float s1 = wave_sum(lane < d0.end - d0.start ? a_raw : 0.f);
      if (lane == 0) aux_sum[row] = s1;
    }
    if (slot == -2) {          // group member (block-uniform: the planner aligns the quadruples and the XCD ranges to 4)
      const int wave = threadIdx.x >> 6;
      if (g == 0) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) grp_red[wave][li + v * LPR] = acc[v];
      }
      if (AUX && lane == 0) grp_aux[wave] = a_sum;
      __syncthreads();
      if (AUX && wave == 0 && lane == 0 && row >= 0) aux_sum[row] = ((grp_aux[0] + grp_aux[1]) + grp_aux[2]) + grp_aux[3];
      if (wave == 0 && g == 0) {
        char* ob = reinterpret_cast<char*>(y + (int64_t)row * ldy);
        const char* sb = reinterpret_cast<const char*>(xs + (int64_t)row * ldx);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          if (!EXACT && li + v * LPR >= d4) continue;
          const int e = li + v * LPR;
          float4 o = f4_add(f4_add(f4_add(grp_red[0][e], grp_red[1][e]), grp_red[2][e]), grp_red[3][e]);
          if (self_coef != 0.0f) o = f4_fma(self_coef, *reinterpret_cast<const float4*>(sb + lo[v]), o);
          if (bias) o = f4_add(o, *reinterpret_cast<const float4*>(bb + lo[v]));
          *reinterpret_cast<float4*>(ob + lo[v]) = o;
        }
      }
      __syncthreads();
    } else if (g == 0 && row >= 0) {
      char* ob = whole ? reinterpret_cast<char*>(y + (int64_t)row * ldy)
                       : reinterpret_cast<char*>(scratch + (int64_t)slot * d4 * 4);
      const char* sb = reinterpret_cast<const char*>(xs + (int64_t)row * ldx);
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        if (!EXACT && li + v * LPR >= d4) continue;
        float4 o = acc[v];
        if (self && !self_in_acc) o = f4_fma(self_coef, *reinterpret_cast<const float4*>(sb + lo[v]), o);
        *reinterpret_cast<float4*>(ob + lo[v]) = o;
      }
    }
    d0 = d1;
    d1 = uniform(dv);
  }
}

template <int LPR, int VPL, int U, bool EXACT, bool ADDR32, bool NOVAL = false>
__global__ __launch_bounds__(256) void spmm_persist_kernel(const int4* __restrict__ items, int32_t n_items,
                                                           const int32_t* __restrict__ xcd_bounds,
                                                           const int32_t* __restrict__ col,
                                                           const float* __restrict__ val,
                                                           const float* __restrict__ x, int64_t ldx,
                                                           float* __restrict__ y, int64_t ldy,
                                                           const float* __restrict__ bias, float self_coef,
                                                           const float* __restrict__ xs,
                                                           float* __restrict__ scratch, int32_t d4, int32_t nnz) {
  spmm_persist_body<LPR, VPL, U, EXACT, ADDR32, false, NOVAL>(items, n_items, xcd_bounds, col, val, x, ldx, y, ldy, bias, self_coef, xs,
                                                              scratch, d4, nnz);
}

// spmm_persist_body with edge values read through a permutation (AUX above): the source-major aggregation of GATConv's backward
// (val / bias / xs / scratch stay kernel arguments although the host passes NULL / x: with constants in their place this
//  instantiation crashes the inliner of ROCm 7.2's clang - updateCGAndAnalysisManagerForPass)
template <int LPR>
__global__ __launch_bounds__(256) void spmm_persist_aux_kernel(const int4* __restrict__ items, int32_t n_items,
                                                               const int32_t* __restrict__ xcd_bounds,
                                                               const int32_t* __restrict__ col, const float* __restrict__ val,
                                                               const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                               int64_t ldy, const float* __restrict__ bias, float self_coef,
                                                               const float* __restrict__ xs, float* __restrict__ scratch, int32_t d4,
                                                               int32_t nnz, const int32_t* __restrict__ perm,
                                                               const float2* __restrict__ aux, float* __restrict__ aux_sum) {
  spmm_persist_body<LPR, 1, 4, true, true, true>(items, n_items, xcd_bounds, col, val, x, ldx, y, ldy, bias, self_coef, xs, scratch, d4,
                                                 nnz, perm, aux, aux_sum);
}

__global__ __launch_bounds__(256) void spmm_fixup_kernel(const int4* __restrict__ split, int32_t n_split,
                                                         const float* __restrict__ scratch,
                                                         const float* __restrict__ x, int64_t ldx,
                                                         float* __restrict__ y, int64_t ldy,
                                                         const float* __restrict__ bias, float self_coef,
                                                         int32_t d4) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_split) return;
  const int4 sp = split[i];
  const int row = sp.x, slot0 = sp.y, n = sp.z;
  for (int vec = lane; vec < d4; vec += kWave) {
    float4 o = f4_zero();
    const float4* sp4 = reinterpret_cast<const float4*>(scratch + (int64_t)slot0 * d4 * 4) + vec;
    int s = 0;
    for (; s + 4 <= n; s += 4) {          // four pieces in flight, added in slot order (same sum as one by one)
      const float4 p0 = sp4[(int64_t)s * d4], p1 = sp4[(int64_t)(s + 1) * d4], p2 = sp4[(int64_t)(s + 2) * d4],
                   p3 = sp4[(int64_t)(s + 3) * d4];
      o = f4_add(f4_add(f4_add(f4_add(o, p0), p1), p2), p3);
    }
    for (; s < n; ++s) o = f4_add(o, sp4[(int64_t)s * d4]);
    if (self_coef != 0.0f) o = f4_fma(self_coef, reinterpret_cast<const float4*>(x + (int64_t)row * ldx)[vec], o);
    if (bias) o = f4_add(o, reinterpret_cast<const float4*>(bias)[vec]);
    reinterpret_cast<float4*>(y + (int64_t)row * ldy)[vec] = o;
  }
}

// Any d / any alignment: one wave per row, lanes stride over columns.
template <bool MEAN>
__global__ __launch_bounds__(256) void spmm_scalar_kernel(const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col,
                                                          const float* __restrict__ val,
                                                          const float* __restrict__ x, int64_t ldx,
                                                          float* __restrict__ y, int64_t ldy,
                                                          const float* __restrict__ bias, float self_coef,
                                                          int32_t n_rows, int32_t x_rows_mod, int32_t d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int start = rowptr[row], end = rowptr[row + 1];
  const int self_row = x_rows_mod > 0 ? row % x_rows_mod : row;
  for (int c0 = 0; c0 < d; c0 += kWave) {
    const int cc = c0 + lane;
    float acc = 0.f;
    for (int k = start; k < end; ++k) {
      const float w = val ? val[k] : 1.0f;
      if (cc < d) acc = fmaf(w, x[(int64_t)col[k] * ldx + cc], acc);
    }
    if (cc < d) {
      if (MEAN) {
        acc = end > start ? acc / (float)(end - start) : 0.f;
      } else {
        if (self_coef != 0.0f) acc = fmaf(self_coef, x[(int64_t)self_row * ldx + cc], acc);
        if (bias) acc += bias[cc];
      }
      y[(int64_t)row * ldy + cc] = acc;
    }
  }
}

__global__ __launch_bounds__(256) void gcn_norm_kernel(const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col, int32_t n_rows,
                                                       float* __restrict__ val) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int start = rowptr[row], end = rowptr[row + 1];
  const float di = 1.0f / sqrtf((float)(end - start));
  for (int k = start + lane; k < end; k += kWave) {
    const int c = col[k];
    const float dj = 1.0f / sqrtf((float)(rowptr[c + 1] - rowptr[c]));
    val[k] = dj * di;
  }
}

template <bool MEAN>
static int launch_spmm(const int32_t* rowptr, const int32_t* col, const float* val, const float* x,
                       int64_t ldx, float* y, int64_t ldy, const float* bias, float self_coef,
                       int32_t n_rows, int32_t x_rows_mod, int32_t d, hipStream_t s) {
  if (n_rows == 0 || d == 0) return GD_OK;
  const dim3 grid((n_rows + 3) / 4), block(256);
  const bool vec_ok = (d % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && aligned16(x) && aligned16(y) &&
                      (!bias || aligned16(bias)) && d <= 1024;
  if (!vec_ok) {
    hipLaunchKernelGGL((spmm_scalar_kernel<MEAN>), grid, block, 0, s, rowptr, col, val, x, ldx, y, ldy, bias,
                       self_coef, n_rows, x_rows_mod, d);
    return launched("spmm_scalar");
  }
  const int d4 = d / 4;
#define GD_SPMM_CASE(LPR, VPL)                                                                              \
  hipLaunchKernelGGL((spmm_vec_kernel<LPR, VPL, MEAN>), grid, block, 0, s, rowptr, col, val, x, ldx, y, ldy, \
                     bias, self_coef, n_rows, x_rows_mod, d4)
  if (d4 <= 1) GD_SPMM_CASE(1, 1);
  else if (d4 <= 2) GD_SPMM_CASE(2, 1);
  else if (d4 <= 4) GD_SPMM_CASE(4, 1);
  else if (d4 <= 8) GD_SPMM_CASE(8, 1);
  else if (d4 <= 16) GD_SPMM_CASE(16, 1);
  else if (d4 <= 32) GD_SPMM_CASE(32, 1);
  else if (d4 <= 64) GD_SPMM_CASE(64, 1);
  else if (d4 <= 128) GD_SPMM_CASE(64, 2);
  else GD_SPMM_CASE(64, 4);
#undef GD_SPMM_CASE
  return launched("spmm_vec");
}

}  // namespace gd

extern "C" int gd_gcn_norm_f32(const int32_t* rowptr, const int32_t* col, int32_t n_rows, float* val,
                               void* stream) {
  GD_REQUIRE(rowptr && col && val, GD_E_NULL, "gd_gcn_norm_f32: null pointer");
  GD_REQUIRE(n_rows >= 0, GD_E_DIM, "gd_gcn_norm_f32: n_rows < 0");
  if (n_rows == 0) return GD_OK;
  hipLaunchKernelGGL(gd::gcn_norm_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, rowptr, col,
                     n_rows, val);
  return gd::launched("gcn_norm");
}

extern "C" int gd_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val, const float* x,
                               int64_t ldx, float* y, int64_t ldy, const float* bias, float self_coef,
                               int32_t n_rows, int32_t d, void* stream) {
  GD_REQUIRE(rowptr && col && x && y, GD_E_NULL, "gd_spmm_csr_f32: null pointer");
  GD_REQUIRE(n_rows >= 0 && d >= 0 && ldx >= d && ldy >= d, GD_E_DIM, "gd_spmm_csr_f32: bad dims n=%d d=%d", n_rows, d);
  GD_REQUIRE(x != y, GD_E_DIM, "gd_spmm_csr_f32: x and y must not alias");
  return gd::launch_spmm<false>(rowptr, col, val, x, ldx, y, ldy, bias, self_coef, n_rows, 0, d,
                                (hipStream_t)stream);
}

extern "C" int gd_rgcn_mean_f32(const int32_t* rowptr, const int32_t* col, const float* x, int64_t ldx, float* y,
                                int64_t ldy, int32_t n_rel, int32_t n_rows, int32_t d, void* stream) {
  GD_REQUIRE(rowptr && col && x && y, GD_E_NULL, "gd_rgcn_mean_f32: null pointer");
  GD_REQUIRE(n_rel >= 0 && n_rows >= 0 && d >= 0 && ldx >= d && ldy >= d, GD_E_DIM, "gd_rgcn_mean_f32: bad dims");
  GD_REQUIRE((int64_t)n_rel * n_rows < (1ll << 31), GD_E_DIM, "gd_rgcn_mean_f32: n_rel*n_rows overflows int32");
  return gd::launch_spmm<true>(rowptr, col, nullptr, x, ldx, y, ldy, nullptr, 0.f, n_rel * n_rows, n_rows, d,
                               (hipStream_t)stream);
}

namespace gd {
// shared launch of the persistent kernel: legacy form (pieces -> scratch, fix-up by the caller) when hubs == nullptr
static int launch_persist(const int32_t* items, int32_t n_items, const int32_t* col,
                          const float* val, const float* x, int64_t ldx, float* y, int64_t ldy, const float* bias,
                          float self_coef, const float* xs, float* scratch, int32_t d, int32_t nnz, int32_t x_rows,
                          const int32_t* xcd_bounds, hipStream_t s) {
  const int d4 = d / 4;
  // grid: a few visits per wave (4 x the resident set of 256 CUs x 8 blocks measured best), a
  // multiple of the 8 XCDs
  int nblk = (n_items + 3) / 4;
  static const int cap = [] {                                       // tuning knob (blocks), read once per process
    const char* e = getenv("GD_SPMM_GRID_CAP");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 8192;
  }();
  if (nblk > cap) nblk = cap;
  nblk = (nblk + 7) / 8 * 8;
  const dim3 grid(nblk), block(256);
  const int4* it = reinterpret_cast<const int4*>(items);
  // 24-bit fast addressing needs: row ids < 2^24, row pitches in bytes < 2^24, x and y smaller than 4 GiB
  const bool addr32 = x_rows > 0 && x_rows <= (1 << 24) && ldx * 4 < (1 << 24) && ldy * 4 < (1 << 24) &&
                      (int64_t)x_rows * ldx * 4 < (1ll << 32) && (int64_t)x_rows * ldy * 4 < (1ll << 32);
#define GD_ITEMS_LAUNCH(LPR, VPL, EXACT, A32)                                                                        \
  hipLaunchKernelGGL((spmm_persist_kernel<LPR, VPL, ((64 / (64 / LPR)) >= 4 ? 4 : (64 / (64 / LPR))), EXACT, A32>), \
                     grid, block, 0, s, it, n_items, xcd_bounds, col, val, x, ldx, y, ldy, bias, self_coef, xs, scratch, \
                     d4, nnz)
#define GD_ITEMS_CASE(LPR, VPL)                                                                                    \
  do {                                                                                                             \
    if (d4 == LPR * VPL) {                                                                                         \
      if (addr32) GD_ITEMS_LAUNCH(LPR, VPL, true, true);                                                           \
      else GD_ITEMS_LAUNCH(LPR, VPL, true, false);                                                                 \
    } else {                                                                                                       \
      if (addr32) GD_ITEMS_LAUNCH(LPR, VPL, false, true);                                                          \
      else GD_ITEMS_LAUNCH(LPR, VPL, false, false);                                                                \
    }                                                                                                              \
  } while (0)
  // LAB (round 6): the 64-float launch in other shapes, chosen per call by GD_SPMM_D64_FORM
  //   8x2u2 / 8x2u4 : eight lanes x two float4 per row (eight neighbours per trip), 2 / 4 trips in flight
  //   + "n" suffix  : the unweighted instantiation when val == NULL (no weight stream, no second crossbar read per trip)
  if (d4 == 16 && addr32) {
    const char* form = getenv("GD_SPMM_D64_FORM");
    if (form && *form) {
      const bool nv = val == nullptr && strchr(form, 'n') != nullptr;
#define GD_LAB_LAUNCH(LPR, VPL, UU, NV)                                                                             \
  hipLaunchKernelGGL((spmm_persist_kernel<LPR, VPL, UU, true, true, NV>), grid, block, 0, s, it, n_items, xcd_bounds, col, val, x, \
                     ldx, y, ldy, bias, self_coef, xs, scratch, d4, nnz)
      if (!strncmp(form, "8x2u2", 5)) { if (nv) GD_LAB_LAUNCH(8, 2, 2, true); else GD_LAB_LAUNCH(8, 2, 2, false); return launched("spmm_persist"); }
      if (!strncmp(form, "8x2u4", 5)) { if (nv) GD_LAB_LAUNCH(8, 2, 4, true); else GD_LAB_LAUNCH(8, 2, 4, false); return launched("spmm_persist"); }
      if (!strncmp(form, "16x1", 4) && nv) { GD_LAB_LAUNCH(16, 1, 4, true); return launched("spmm_persist"); }
#undef GD_LAB_LAUNCH
    }
  }
  if (d4 <= 1) GD_ITEMS_CASE(1, 1);
  else if (d4 <= 2) GD_ITEMS_CASE(2, 1);
  else if (d4 <= 4) GD_ITEMS_CASE(4, 1);
  else if (d4 <= 8) GD_ITEMS_CASE(8, 1);
  else if (d4 <= 16) GD_ITEMS_CASE(16, 1);
  else if (d4 <= 32) GD_ITEMS_CASE(32, 1);
  else if (d4 <= 64) GD_ITEMS_CASE(64, 1);
  else if (d4 <= 128) GD_ITEMS_CASE(64, 2);
  else GD_ITEMS_CASE(64, 4);
#undef GD_ITEMS_CASE
#undef GD_ITEMS_LAUNCH
  return launched("spmm_persist");
}
}  // namespace gd

extern "C" int gd_spmm_csr_balanced_f32(const int32_t* items, int32_t n_items, const int32_t* split, int32_t n_split,
                                        const int32_t* col, const float* val, const float* x, int64_t ldx, float* y,
                                        int64_t ldy, const float* bias, float self_coef, const float* x_self,
                                        float* scratch, int32_t d, int32_t nnz, int32_t x_rows,
                                        const int32_t* xcd_bounds, void* stream) {
  using namespace gd;
  GD_REQUIRE(items && col && x && y, GD_E_NULL, "gd_spmm_csr_balanced_f32: null pointer");
  GD_REQUIRE(n_split == 0 || (split && scratch), GD_E_NULL, "gd_spmm_csr_balanced_f32: split rows need scratch");
  GD_REQUIRE(n_items >= 0 && n_split >= 0 && d > 0 && d % 4 == 0 && d <= 1024 && ldx % 4 == 0 && ldy % 4 == 0, GD_E_DIM,
             "gd_spmm_csr_balanced_f32: d=%d must be a multiple of 4 (<=1024) with 16-byte row strides", d);
  GD_REQUIRE(aligned16(x) && aligned16(y) && aligned16(items) && (!bias || aligned16(bias)) &&
                 (!scratch || aligned16(scratch)) && (!split || aligned16(split)), GD_E_ALIGN,
             "gd_spmm_csr_balanced_f32: unaligned pointer");
  GD_REQUIRE(x != y, GD_E_DIM, "gd_spmm_csr_balanced_f32: x and y must not alias");
  if (n_items == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const int d4 = d / 4;
  const float* xs = x_self ? x_self : x;            // rows of the self / residual term (same pitch as x)
  GD_REQUIRE(aligned16(xs) && xs != y, GD_E_ALIGN, "gd_spmm_csr_balanced_f32: bad x_self");
  // (a plan without split rows needs no scratch rows; the kernel tells the two forms apart by this pointer - see its prologue -
  //  so the piece form always hands it one: never dereferenced without split rows)
  float* scratch_arg = scratch ? scratch : reinterpret_cast<float*>(const_cast<int32_t*>(items));
  int rc = launch_persist(items, n_items, col, val, x, ldx, y, ldy, bias, self_coef, xs, scratch_arg, d, nnz, x_rows,
                          xcd_bounds, s);
  if (rc || n_split == 0) return rc;
  hipLaunchKernelGGL(spmm_fixup_kernel, dim3((n_split + 3) / 4), dim3(256), 0, s, reinterpret_cast<const int4*>(split),
                     n_split, scratch, xs, ldx, y, ldy, bias, self_coef, d4);
  return launched("spmm_fixup");
}

extern "C" int gd_spmm_csr_onepass_f32(const int32_t* items, int32_t n_items, const int32_t* col, const float* val,
                                       const float* x, int64_t ldx, float* y, int64_t ldy, const float* bias,
                                       float self_coef, const float* x_self, int32_t d, int32_t nnz, int32_t x_rows,
                                       const int32_t* xcd_bounds, void* stream) {
  using namespace gd;
  GD_REQUIRE(col && x && y && xcd_bounds && (items || n_items == 0), GD_E_NULL, "gd_spmm_csr_onepass_f32: null pointer");
  GD_REQUIRE(n_items >= 0 && n_items % 4 == 0 && d > 0 && d % 4 == 0 && d <= 1024 && ldx % 4 == 0 && ldy % 4 == 0, GD_E_DIM,
             "gd_spmm_csr_onepass_f32: d=%d must be a multiple of 4 (<=1024) with 16-byte row strides, n_items a multiple of 4", d);
  GD_REQUIRE(aligned16(x) && aligned16(y) && aligned16(items) && (!bias || aligned16(bias)), GD_E_ALIGN,
             "gd_spmm_csr_onepass_f32: unaligned pointer");
  GD_REQUIRE(x != y, GD_E_DIM, "gd_spmm_csr_onepass_f32: x and y must not alias");
  if (n_items == 0) return GD_OK;
  const float* xs = x_self ? x_self : x;
  GD_REQUIRE(aligned16(xs) && xs != y, GD_E_ALIGN, "gd_spmm_csr_onepass_f32: bad x_self");
  return launch_persist(items, n_items, col, val, x, ldx, y, ldy, bias, self_coef, xs, nullptr, d, nnz, x_rows,
                        xcd_bounds, (hipStream_t)stream);
}

extern "C" int gd_spmm_csr_onepass_aux_f32(const int32_t* items, int32_t n_items, const int32_t* col, const int32_t* perm,
                                           const float* aux, const float* x, int64_t ldx, float* y, int64_t ldy, float* aux_sum,
                                           int32_t d, int32_t nnz, int32_t x_rows, const int32_t* xcd_bounds, void* stream) {
  using namespace gd;
  GD_REQUIRE(col && perm && aux && x && y && aux_sum && xcd_bounds && (items || n_items == 0), GD_E_NULL,
             "gd_spmm_csr_onepass_aux_f32: null pointer");
  GD_REQUIRE(n_items >= 0 && n_items % 4 == 0 && (d == 64 || d == 128) && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= d && ldy >= d, GD_E_DIM,
             "gd_spmm_csr_onepass_aux_f32: d must be 64 or 128 (got %d), 16-byte row strides, n_items a multiple of 4", d);
  GD_REQUIRE(aligned16(x) && aligned16(y) && aligned16(items) && (reinterpret_cast<uintptr_t>(aux) & 7u) == 0, GD_E_ALIGN,
             "gd_spmm_csr_onepass_aux_f32: unaligned pointer");
  GD_REQUIRE(x != y, GD_E_DIM, "gd_spmm_csr_onepass_aux_f32: x and y must not alias");
  GD_REQUIRE(x_rows > 0 && x_rows <= (1 << 24) && ldx * 4 < (1 << 24) && ldy * 4 < (1 << 24) &&
                 (int64_t)x_rows * ldx * 4 < (1ll << 32) && (int64_t)x_rows * ldy * 4 < (1ll << 32), GD_E_DIM,
             "gd_spmm_csr_onepass_aux_f32: x and y must be smaller than 4 GiB with row ids and pitches below 2^24");
  if (n_items == 0) return GD_OK;
  int nblk = (n_items + 3) / 4;
  if (nblk > 8192) nblk = 8192;
  nblk = (nblk + 7) / 8 * 8;
  const int4* it = reinterpret_cast<const int4*>(items);
  const float2* a2 = reinterpret_cast<const float2*>(aux);
  if (d == 64)
    hipLaunchKernelGGL((spmm_persist_aux_kernel<16>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, it, n_items, xcd_bounds, col, (const float*)nullptr,
                       x, ldx, y, ldy, (const float*)nullptr, 0.0f, x, (float*)nullptr, d / 4, nnz, perm, a2, aux_sum);
  else
    hipLaunchKernelGGL((spmm_persist_aux_kernel<32>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, it, n_items, xcd_bounds, col, (const float*)nullptr,
                       x, ldx, y, ldy, (const float*)nullptr, 0.0f, x, (float*)nullptr, d / 4, nnz, perm, a2, aux_sum);
  return launched("spmm_persist_aux");
}
