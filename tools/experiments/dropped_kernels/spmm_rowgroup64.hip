// gd_spmm_csr_rowgroup_f32 (round 4): cut out of csrc/spmm.hip in round 5 - measured no faster than the item kernel, nobody defaulted it.
// Needs the helpers of csrc/common.h; planner: SplitPlan.rowgroup in the round-4 tree (git show 4ff847f:gnndelete_amd/graph.py).
#include "common.h"
namespace gd {
// ---------------------------------------------------------------------------------------------
// Row-per-lane-group form for 64-float rows (the layer-2 aggregations of the step, forward and transposed).
// The item kernel above deals an item's edges to its four lane groups and tree-sums the groups; at 64 floats a row is one
// lane group wide, the average row has 8 in-edges, and the issue-side counters of rounds 2-3 show that kernel saturating
// the instruction issue of its SIMDs (142 VALU + 106 SALU instructions per two-row visit; waves "executing" 17 % x 7
// resident waves) while the fabric idles at 4.3-4.8 of its 6.1 TB/s.  Here every lane group (16 lanes x 16 bytes) OWNS one
// row: it walks that row's in-edges in order into ONE accumulator - no dealing of edges, no select per gathered row, no
// cross-group tree, one 1 KB store for four rows.  items[4 i + g] = {row, start, end, meta} of lane group g of item i:
//   meta of group 0 = kind << 24 | trips (the longest group's edge count, scalar trip control);
//   kind 0  four rows (the planner packs rows of similar length; a short last pack repeats its last row - the repeat
//           recomputes and rewrites that row with identical values);
//   kind 1  ONE row of 65 .. 512 in-edges dealt to the four groups as contiguous shares, summed across the groups;
//   kind 2  one of four consecutive, 4-aligned items that the four waves of a block process together: a heavier row in
//           sixteen shares, the waves' sums meet in LDS (as the hub groups of the item kernel);   kind 3  padding.
// MEASURED (round 4, bench step): no faster than the item kernel - spmm2 / spmm2_t 66.7 / 64.0 us against 68.2 / 60.2 in step,
// PMC traffic 334.7 MB against 288.6 MB per launch (moved at 5.0-5.2 TB/s where the item kernel moves its bytes at 4.3-4.8);
// a two-rows-per-item variant (two lane groups per row) moved the same 336 MB.  OPT-IN (GD_SPMM_ROWGROUP=1), not the default.
// Sums are sequential over a row's (sorted) in-edges - another association than the item kernel's (equal to fp32
// rounding), fixed from run to run.  Same persistent XCD sweep and prefetch pipeline as above (descriptor two visits
// ahead, the first 16 (col, val) of every group one visit ahead: one dependent round trip per visit).
template <int U>
__global__ __launch_bounds__(256) void spmm_rowgroup64_kernel(const int4* __restrict__ items, int32_t n_items,
                                                              const int32_t* __restrict__ xcd_bounds,
                                                              const int32_t* __restrict__ col, const float* __restrict__ val,
                                                              const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                              int64_t ldy, const float* __restrict__ bias, float self_coef,
                                                              const float* __restrict__ xs, int32_t nnz) {
  constexpr int kXcd = 8;
  const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x % kXcd;
  const int stride = (gridDim.x / kXcd) * 4;                        // waves per XCD
  const int wx = (blockIdx.x / kXcd) * 4 + wave;
  const int i0 = xcd_bounds[xcd], i1 = xcd_bounds[xcd + 1];
  int i = i0 + wx;
  if (i >= i1) return;
  const uint32_t lo16 = 16u * (uint32_t)li;
  const uint32_t pitch_b = (uint32_t)ldx * 4u;
  const char* xb = reinterpret_cast<const char*>(x);
  const int lane_base4 = 4 * (g * 16);                              // byte address of the group's lane 0 for ds_bpermute
  __shared__ float4 grp_red[4][16];

  int4 d0 = items[(int64_t)i * 4 + g], d1 = items[(int64_t)min(i + stride, i1 - 1) * 4 + g];
  int kk = min(d0.y + li, nnz - 1);
  int c = col[kk];
  float w = val ? val[kk] : 1.0f;
  for (; i < i1; i += stride) {
    const int4 d2 = items[(int64_t)min(i + 2 * stride, i1 - 1) * 4 + g];
    const int row = d0.x, start = d0.y, len = d0.z - d0.y;
    const int meta = __builtin_amdgcn_readfirstlane(d0.w);
    const int kind = meta >> 24, trips = meta & 0xffffff;
    int c_cur = c;
    float w_raw = w;
    kk = min(d1.y + li, nnz - 1);                                   // next visit's indices
    c = col[kk];
    w = val ? val[kk] : 1.0f;
    float4 acc = f4_zero();
    if (kind == 0) {                                                // the row's constant terms ride with the gathers
      if (bias) acc = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bias) + lo16);
      if (self_coef != 0.0f)
        acc = f4_fma(self_coef, *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(xs + (int64_t)row * ldx) + lo16), acc);
    }
    for (int t0 = 0; t0 < trips; t0 += 16) {                        // 16-edge chunks of the groups' rows
      const int rem = len - t0;                                     // this group's edges from t0 on (<= 0: done)
      const float w_cur = li < rem ? w_raw : 0.f;
      const int last4 = lane_base4 + 4 * max(min(rem, 16) - 1, 0);  // idle trips re-read the group's last neighbour with weight 0
      const int tc = min(16, trips - t0);
      auto fetch = [&](int j, float4& xv, float& wj) {
        const int a4 = min(lane_base4 + 4 * j, last4);
        const int cs = __builtin_amdgcn_ds_bpermute(a4, c_cur);
        wj = __int_as_float(__builtin_amdgcn_ds_bpermute(lane_base4 + 4 * j, __float_as_int(w_cur)));
        xv = *reinterpret_cast<const float4*>(xb + (__umul24((uint32_t)cs, pitch_b) + lo16));
      };
      int t = 0;
      for (; t + U <= tc; t += U) {
        float4 xv[U];
        float wj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(t + u, xv[u], wj[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) acc = f4_fma(wj[u], xv[u], acc);
      }
      if (t < tc) {
        float4 xv[U - 1];
        float wj[U - 1];
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
          if (t + u < tc) fetch(t + u, xv[u], wj[u]);
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
          if (t + u < tc) acc = f4_fma(wj[u], xv[u], acc);
      }
      if (t0 + 16 < trips) {                                        // rows above 16 in-edges: next chunk fetched in place
        const int k2 = min(start + t0 + 16 + li, nnz - 1);
        c_cur = col[k2];
        w_raw = val ? val[k2] : 1.0f;
      }
    }
    if (kind == 0) {
      *reinterpret_cast<float4*>(reinterpret_cast<char*>(y + (int64_t)row * ldy) + lo16) = acc;
    } else if (kind != 3) {
      acc = make_float4(xor16_sum(acc.x), xor16_sum(acc.y), xor16_sum(acc.z), xor16_sum(acc.w));
      acc = make_float4(xor32_sum(acc.x), xor32_sum(acc.y), xor32_sum(acc.z), xor32_sum(acc.w));
      if (kind == 2) {            // (block-uniform: the planner aligns these quadruples and the XCD ranges to 4)
        if (g == 0) grp_red[wave][li] = acc;
        __syncthreads();
        if (wave == 0) acc = f4_add(f4_add(f4_add(grp_red[0][li], grp_red[1][li]), grp_red[2][li]), grp_red[3][li]);
      }
      if (g == 0 && (kind == 1 || wave == 0)) {
        if (self_coef != 0.0f)
          acc = f4_fma(self_coef, *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(xs + (int64_t)row * ldx) + lo16), acc);
        if (bias) acc = f4_add(acc, *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bias) + lo16));
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(y + (int64_t)row * ldy) + lo16) = acc;
      }
      if (kind == 2) __syncthreads();
    }
    d0 = d1;
    d1 = d2;
  }
}

}  // namespace gd

extern "C" int gd_spmm_csr_rowgroup_f32(const int32_t* items, int32_t n_items, const int32_t* col, const float* val,
                                        const float* x, int64_t ldx, float* y, int64_t ldy, const float* bias,
                                        float self_coef, const float* x_self, int32_t d, int32_t nnz, int32_t x_rows,
                                        const int32_t* xcd_bounds, void* stream) {
  using namespace gd;
  GD_REQUIRE(col && x && y && xcd_bounds && (items || n_items == 0), GD_E_NULL, "gd_spmm_csr_rowgroup_f32: null pointer");
  GD_REQUIRE(n_items >= 0 && n_items % 4 == 0 && d == 64 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= d && ldy >= d, GD_E_DIM,
             "gd_spmm_csr_rowgroup_f32: d must be 64 (got %d), 16-byte row strides, n_items a multiple of 4", d);
  GD_REQUIRE(aligned16(x) && aligned16(y) && aligned16(items) && (!bias || aligned16(bias)), GD_E_ALIGN,
             "gd_spmm_csr_rowgroup_f32: unaligned pointer");
  GD_REQUIRE(x != y, GD_E_DIM, "gd_spmm_csr_rowgroup_f32: x and y must not alias");
  GD_REQUIRE(x_rows > 0 && x_rows <= (1 << 24) && ldx * 4 < (1 << 24) && (int64_t)x_rows * ldx * 4 < (1ll << 32), GD_E_DIM,
             "gd_spmm_csr_rowgroup_f32: x must be smaller than 4 GiB with row ids and pitches below 2^24");
  if (n_items == 0) return GD_OK;
  const float* xs = x_self ? x_self : x;
  GD_REQUIRE(aligned16(xs) && xs != y, GD_E_ALIGN, "gd_spmm_csr_rowgroup_f32: bad x_self");
  int nblk = (n_items + 3) / 4;
  // (measured: 512 .. 4096 persistent blocks are all slower than handing every wave one or two visits - 0.77 .. 0.62 ms
  //  per step against 0.60: a wave's visits are serial round trips, fresh waves overlap them)
  static const int cap = [] { const char* e = getenv("GD_SPMM_ROWGROUP_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 8192; }();
  if (nblk > cap) nblk = cap;
  nblk = (nblk + 7) / 8 * 8;
  static const int u_sel = [] { const char* e = getenv("GD_SPMM_ROWGROUP_U"); return e ? atoi(e) : 4; }();
  if (u_sel == 8)
    hipLaunchKernelGGL((spmm_rowgroup64_kernel<8>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const int4*>(items),
                       n_items, xcd_bounds, col, val, x, ldx, y, ldy, bias, self_coef, xs, nnz);
  else
    hipLaunchKernelGGL((spmm_rowgroup64_kernel<4>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const int4*>(items),
                     n_items, xcd_bounds, col, val, x, ldx, y, ldy, bias, self_coef, xs, nnz);
  return launched("spmm_rowgroup64");
}
