// Weight gradient of the first Del operator with the layer's loss formed in the fetch, 128 x 128, output-stationary:
//
//     dW = sum_s a[ia(s), :]^T g(s, :),     g(s, :) = coef_u (z[iz(s), :] - tbar_u) + g_add[iz(s), :]      (u = loss slot of s, or none)
//
// The LDS-tile form (rows_gemm.hip: rows_wgrad_mfma_kernel, blocks of 8 waves alternating fetch and MFMA phases over
// double-buffered 32-row tiles) moves its 366 MB at 4.2-4.7 TB/s.  Here ONE wave per SIMD keeps the WHOLE 128 x 128 sum in its
// accumulator registers for all the rows it is given - as two 128 x 64 halves held by a PAIR of waves (32 tiles of
// v_mfma_f32_16x16x4_f32 = 128 registers each) that walk the same rows; K = rows:
//   unit = 16 rows.  lane (r = lane & 15, kq = lane >> 4) fetches the 128-byte slices kq of row r of a, z, tbar and g_add (16-byte
//   loads), forms g (and the loss sums) and writes a and g sample-major into two wave-private 16 x 144 LDS tiles; read back
//   feature-major they are the MFMA operands:  A[i][k] = a[row 4 sp + kq][16 ta + i],  B[k][j] = g[row 4 sp + kq][16 tb + j],
//   dW[16 ta + 4 kq + v][16 tb + j] += ...   (4 x 8 x 4 = 128 matrix instructions per unit and wave).
//   The next unit's rows are requested as soon as this unit's are in the tiles (row ids / loss slots two units ahead) and
//   land under the 256 matrix instructions; no block barrier in the loop.
// Rows past the end are clamped to the last row with their g and loss contribution zeroed.  A pair's two halves are ONE
// partial matrix (no reduction across waves), counted and laid out like the LDS-tile form's (gd_rows_gemm_wgrad_blocks(n_sel)
// partials, reduced by gd_step_tail_f32 / gd_rows_gemm_wgrad_reduce_f32); slots beyond the pairs are written as zeros.
#include <stdlib.h>

#include "common.h"

namespace gd {

typedef float f32x4g __attribute__((ext_vector_type(4)));

struct WgradLossWs {
  const int32_t* slot; const float* tm; const float* coef; const float* cnt_signed; float* partials;
};

// Two waves (a PAIR) share every row unit: each keeps one 128 x 64 half of the sum (columns 64 hb .. 64 hb + 63 of g: 32 tiles
// = 128 accumulator registers - the whole 128 x 128 sum in one wave leaves no room for a unit of rows in flight and spilled).
// Both fetch the a rows (the second read hits the cache), each its half of z / tbar / g_add; a pair's two halves are ONE
// partial matrix: no reduction across waves at all.
template <bool HAS_ADD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rows_wgrad_loss_ws_kernel(
    const float* __restrict__ a, int64_t ld_a, const int32_t* __restrict__ a_idx, const float* __restrict__ z, int64_t ld_z,
    const int32_t* __restrict__ z_idx, const float* __restrict__ g_add, int32_t n_sel, WgradLossWs loss,
    float* __restrict__ partials, int32_t n_part) {
  constexpr int D = 128, PA = 144, PG = 80;                       // tile pitches (rows 16 banks apart)
  extern __shared__ __attribute__((aligned(16))) float wl[];      // 4 waves x (16 x PA + 16 x PG)
  __shared__ float lred[2][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kq = lane >> 4;
  const int hb = wave & 1, pair = blockIdx.x * 2 + (wave >> 1), n_pairs = gridDim.x * 2;
  float* const ta_ = wl + wave * (16 * PA + 16 * PG);
  float* const tg_ = ta_ + 16 * PA;
  const int n_units = (n_sel + 15) >> 4;
  const int u_lo = (int)((int64_t)n_units * pair / n_pairs), u_hi = (int)((int64_t)n_units * (pair + 1) / n_pairs);
  auto slot_of = [&](int u) -> int { return min(min(u, n_units - 1) * 16 + r, n_sel - 1); };

  f32x4g gacc[32];                                                // dW[16 ta + 4 kq + v][64 hb + 16 tb + r] in gacc[ta * 4 + tb][v]
#pragma unroll
  for (int i = 0; i < 32; ++i) gacc[i] = f32x4g{0.f, 0.f, 0.f, 0.f};
  float ls0 = 0.f, ls1 = 0.f;

  // descriptors two units ahead, rows one unit ahead (in place: a unit's registers are reloaded once they are in the tiles)
  int32_t ra_n = a_idx[slot_of(u_lo)], rz_n = z_idx[slot_of(u_lo)], ls_n = loss.slot[slot_of(u_lo)];
  int32_t ra_nn = a_idx[slot_of(u_lo + 1)], rz_nn = z_idx[slot_of(u_lo + 1)], ls_nn = loss.slot[slot_of(u_lo + 1)];
  float4 xa0, xa1, xa2, xa3, xa4, xa5, xa6, xa7, xz0, xz1, xz2, xz3, xt0, xt1, xt2, xt3;
  float4 xd0 = f4_zero(), xd1 = f4_zero(), xd2 = f4_zero(), xd3 = f4_zero();
  float cf_raw, cn_raw;
  int32_t ls_cur;
  // (a macro, not a lambda: with the row arrays captured by reference the 8-vector array stayed in scratch memory)
#define GD_WG_FETCH(ra, rz, u)                                                                                          \
  do {                                                                                                                  \
    const float4* pa = reinterpret_cast<const float4*>(a + (int64_t)(ra) * ld_a + kq * 32);                             \
    const float4* pz = reinterpret_cast<const float4*>(z + (int64_t)(rz) * ld_z + 64 * hb + kq * 16);                   \
    const int uc = max((u), 0);                                                                                         \
    const float4* pt = reinterpret_cast<const float4*>(loss.tm + (int64_t)uc * D + 64 * hb + kq * 16);                  \
    xa0 = pa[0]; xa1 = pa[1]; xa2 = pa[2]; xa3 = pa[3]; xa4 = pa[4]; xa5 = pa[5]; xa6 = pa[6]; xa7 = pa[7];             \
    xz0 = pz[0]; xz1 = pz[1]; xz2 = pz[2]; xz3 = pz[3];                                                                 \
    xt0 = pt[0]; xt1 = pt[1]; xt2 = pt[2]; xt3 = pt[3];                                                                 \
    if (HAS_ADD) {                                                                                                      \
      const float4* pd = reinterpret_cast<const float4*>(g_add + (int64_t)(rz) * ld_z + 64 * hb + kq * 16);             \
      xd0 = pd[0]; xd1 = pd[1]; xd2 = pd[2]; xd3 = pd[3];                                                               \
    }                                                                                                                   \
    cf_raw = loss.coef[uc];                                                                                             \
    cn_raw = loss.cnt_signed[uc];                                                                                       \
    ls_cur = (u);                                                                                                       \
  } while (0)
  if (u_lo < u_hi) {
    GD_WG_FETCH(ra_n, rz_n, ls_n);
    for (int u = u_lo; u < u_hi; ++u) {
      const float livef = min(u, n_units - 1) * 16 + r < n_sel ? 1.f : 0.f;
      ra_n = ra_nn; rz_n = rz_nn; ls_n = ls_nn;
      ra_nn = a_idx[slot_of(u + 2)];                               // (both index lists required: a load under a branch makes
      rz_nn = z_idx[slot_of(u + 2)];                               //  every wait of the loop a vmcnt(0))
      ls_nn = loss.slot[slot_of(u + 2)];
      // ---- this unit's operands move to their own registers and the NEXT unit's rows are requested at once: a whole unit
      // (~2.5 us) to land - requested after the tile writes they had only the product (1.9 us) and every unit began with a wait
      const float4 xz[4] = {xz0, xz1, xz2, xz3}, xt[4] = {xt0, xt1, xt2, xt3}, xd[4] = {xd0, xd1, xd2, xd3};
      const float4 xa[8] = {xa0, xa1, xa2, xa3, xa4, xa5, xa6, xa7};
      const float cf = ls_cur >= 0 ? cf_raw * livef : 0.f, cn = ls_cur >= 0 ? cn_raw * livef : 0.f;
      __builtin_amdgcn_sched_barrier(0);
      GD_WG_FETCH(ra_n, rz_n, ls_n);
      __builtin_amdgcn_sched_barrier(0);
      // ---- this wave's half of g and of the loss sums; a and g into the tiles (sample-major)
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 df = make_float4(xz[i].x - xt[i].x, xz[i].y - xt[i].y, xz[i].z - xt[i].z, xz[i].w - xt[i].w);
        sq = fmaf(df.x, df.x, sq); sq = fmaf(df.y, df.y, sq); sq = fmaf(df.z, df.z, sq); sq = fmaf(df.w, df.w, sq);
        float4 gv = make_float4(cf * df.x, cf * df.y, cf * df.z, cf * df.w);
        if (HAS_ADD) gv = make_float4(fmaf(livef, xd[i].x, gv.x), fmaf(livef, xd[i].y, gv.y), fmaf(livef, xd[i].z, gv.z), fmaf(livef, xd[i].w, gv.w));
        *reinterpret_cast<float4*>(tg_ + r * PG + kq * 16 + 4 * i) = gv;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(ta_ + r * PA + kq * 32 + 4 * i) = xa[i];
      if (cn >= 0.f) ls0 = fmaf(cn, sq, ls0); else ls1 = fmaf(-cn, sq, ls1);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // wave-private tiles: DS operations execute in issue order
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- dW half += a^T g over the unit's 16 rows, operands feature-major out of the tiles (four rows per pass)
      // (operands of pass sp + 1 are requested before the matrix instructions of pass sp are issued)
      float av[2][8], bv[2][4];
#pragma unroll
      for (int t = 0; t < 8; ++t) av[0][t] = ta_[kq * PA + 16 * t + r];
#pragma unroll
      for (int t = 0; t < 4; ++t) bv[0][t] = tg_[kq * PG + 16 * t + r];
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        if (sp < 3) {
#pragma unroll
          for (int t = 0; t < 8; ++t) av[(sp + 1) & 1][t] = ta_[(4 * (sp + 1) + kq) * PA + 16 * t + r];
#pragma unroll
          for (int t = 0; t < 4; ++t) bv[(sp + 1) & 1][t] = tg_[(4 * (sp + 1) + kq) * PG + 16 * t + r];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ta = 0; ta < 8; ++ta)
#pragma unroll
          for (int tb = 0; tb < 4; ++tb)
            gacc[ta * 4 + tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sp & 1][ta], bv[sp & 1][tb], gacc[ta * 4 + tb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // (the tiles are read before the next unit overwrites them)
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef GD_WG_FETCH
  // ---- a pair's two halves are one partial matrix; slots beyond the pairs: zeros
  if (pair < n_part) {
    float* const out = partials + (int64_t)pair * D * D + 64 * hb;
#pragma unroll
    for (int ta = 0; ta < 8; ++ta)
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int v = 0; v < 4; ++v) out[(16 * ta + 4 * kq + v) * D + 16 * tb + r] = gacc[ta * 4 + tb][v];
  }
  for (int slot = n_pairs + blockIdx.x; slot < n_part; slot += gridDim.x) {
    float4* const out = reinterpret_cast<float4*>(partials + (int64_t)slot * D * D);
    for (int e = tid; e < D * D / 4; e += 256) out[e] = f4_zero();
  }
  ls0 = wave_sum(ls0);
  ls1 = wave_sum(ls1);
  if (lane == 0) { lred[0][wave] = ls0; lred[1][wave] = ls1; }
  __syncthreads();
  if (tid < 2) {                                                  // one pair of loss sums per partial slot (= per wave pair)
    const int slot = blockIdx.x * 2 + tid;
    if (slot < n_part) { loss.partials[2 * slot] = lred[0][2 * tid] + lred[0][2 * tid + 1]; loss.partials[2 * slot + 1] = lred[1][2 * tid] + lred[1][2 * tid + 1]; }
  }
  if (tid == 0)
    for (int slot = n_pairs + blockIdx.x; slot < n_part; slot += gridDim.x) { loss.partials[2 * slot] = 0.f; loss.partials[2 * slot + 1] = 0.f; }
}

// -> GD_OK / error when this form took the call, 1 when it does not cover it (128 x 128, loss form, >= 65,536 rows,
// fp32 products, 16-byte aligned rows)
int rows_wgrad_loss_ws_try(const float* a, int64_t ld_a, const int32_t* a_idx, const float* z, int64_t ld_z, const int32_t* z_idx,
                           const int32_t* loss_slot, const float* tm, const float* coef, const float* cnt_signed, const float* g_add,
                           int32_t n_sel, int32_t d_a, int32_t d_b, float* partials, float* loss_partials, int32_t n_part, void* stream) {
  // OPT-IN (GD_WGRAD_WS=1): measured SLOWER than the LDS-tile form in the step (93 against 85 us, 420 against 403 MB of
  // traffic: the a rows are fetched by both waves of a pair, and a lone wave per SIMD serialises its fetch / tile / product
  // phases) - kept parity-tested as the record of the experiment (profiles/NOTES.md, round 4)
  static const bool on = [] { const char* e = getenv("GD_WGRAD_WS"); return e && atoi(e) == 1; }();
  if (!on || d_a != 128 || d_b != 128 || n_sel < 65536 || n_part < 1 || !a_idx || !z_idx) return 1;
  if (!aligned16(a) || !aligned16(z) || !aligned16(tm) || (g_add && !aligned16(g_add)) || ld_a % 4 || ld_z % 4 || !aligned16(partials)) return 1;
  const int cus = ws_cu_count();
  const int grid = 2 * cus <= n_part ? cus : n_part / 2;           // two wave pairs = two partial matrices per block
  if (grid < 1) return 1;
  constexpr int kLds = 4 * (16 * 144 + 16 * 80) * 4;
  static const hipError_t once1 = hipFuncSetAttribute((const void*)rows_wgrad_loss_ws_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  static const hipError_t once0 = hipFuncSetAttribute((const void*)rows_wgrad_loss_ws_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  GD_REQUIRE(once1 == hipSuccess && once0 == hipSuccess, -(int)(once1 != hipSuccess ? once1 : once0),
             "gd_rows_gemm_wgrad_loss_f32: cannot raise the LDS limit of the output-stationary kernel");
  const WgradLossWs loss{loss_slot, tm, coef, cnt_signed, loss_partials};
  if (g_add)
    hipLaunchKernelGGL(rows_wgrad_loss_ws_kernel<true>, dim3(grid), dim3(256), kLds, (hipStream_t)stream, a, ld_a, a_idx, z, ld_z, z_idx, g_add,
                       n_sel, loss, partials, n_part);
  else
    hipLaunchKernelGGL(rows_wgrad_loss_ws_kernel<false>, dim3(grid), dim3(256), kLds, (hipStream_t)stream, a, ld_a, a_idx, z, ld_z, z_idx, g_add,
                       n_sel, loss, partials, n_part);
  return launched("rows_wgrad_loss_ws");
}

}  // namespace gd
