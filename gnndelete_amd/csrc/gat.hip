// GAT (heads = 1) attention-scored aggregation: fused edge-score / segment-softmax / weighted
// gather in one kernel (the reference runs ~8 separate gather / scatter kernels per layer), and
// its backward as two kernels (target-major pass for the softmax Jacobian, source-major pass for
// the message gradient).  One wave per CSR row, LPR lanes x float4 per feature row, G = 64/LPR
// neighbours in flight; scores of up to 64 in-edges live in one register per lane.
#include <math.h>

#include "common.h"

namespace gd {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float leaky(float s, float slope) { return s > 0.f ? s : slope * s; }

template <int LPR, int VPL>
__global__ __launch_bounds__(256) void gat_fwd_kernel(const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ col,
                                                      const float* __restrict__ a_src, const float* __restrict__ a_dst,
                                                      const float* __restrict__ h, int64_t ldh, float* __restrict__ y,
                                                      int64_t ldy, const float* __restrict__ bias,
                                                      float* __restrict__ alpha_out, float slope, int32_t n_rows,
                                                      int32_t d4) {
  constexpr int G = kWave / LPR;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int g = lane / LPR, li = lane % LPR;
  const int start = rowptr[row], end = rowptr[row + 1];
  const float ad = a_dst[row];

  // pass 1: segment max
  float mx = -INFINITY;
  for (int k = start + lane; k < end; k += kWave) mx = fmaxf(mx, leaky(a_src[col[k]] + ad, slope));
  mx = wave_max(mx);
  // pass 2: segment sum of exp
  float sm = 0.f;
  for (int k = start + lane; k < end; k += kWave) sm += expf(leaky(a_src[col[k]] + ad, slope) - mx);
  sm = wave_sum(sm);
  const float inv = 1.0f / (sm + 1e-16f);

  // pass 3: weighted gather
  float4 acc[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = f4_zero();
  for (int base = start; base < end; base += kWave) {
    const int k = base + lane;
    const bool live = k < end;
    const int c = live ? col[k] : 0;
    const float w = live ? expf(leaky(a_src[c] + ad, slope) - mx) * inv : 0.f;
    if (alpha_out && live) alpha_out[k] = w;
    const int cnt = min(kWave, end - base);
    const int trips = (cnt + G - 1) / G;

    for (int it = 0; it < trips; ++it) {
      const int j = it * G + g;
      const int cj = __shfl(c, j);
      const float wj = __shfl(w, j);
      const float4* hr = reinterpret_cast<const float4*>(h + (int64_t)cj * ldh);
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int vec = li + v * LPR;
        if (vec < d4) acc[v] = f4_fma(wj, hr[vec], acc[v]);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VPL; ++v)
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) acc[v] = f4_add(acc[v], f4_shfl_xor(acc[v], off));
  if (g != 0) return;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int vec = li + v * LPR;
    if (vec >= d4) continue;
    float4 o = acc[v];
    if (bias) o = f4_add(o, reinterpret_cast<const float4*>(bias)[vec]);
    reinterpret_cast<float4*>(y + (int64_t)row * ldy)[vec] = o;
  }
}

// backward, target-major: d_alpha_e = <dy_i, h_j>; de_e = alpha_e (d_alpha_e - sum alpha d_alpha) * leaky'
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void gat_bwd_row_kernel(const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col,
                                                          const float* __restrict__ alpha,
                                                          const float* __restrict__ a_src,
                                                          const float* __restrict__ a_dst, const float* __restrict__ h,
                                                          int64_t ldh, const float* __restrict__ dy, int64_t lddy,
                                                          float* __restrict__ de, float* __restrict__ da_dst,
                                                          float slope, int32_t n_rows, int32_t d4) {
  constexpr int G = kWave / LPR;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int g = lane / LPR, li = lane % LPR;
  const int start = rowptr[row], end = rowptr[row + 1];
  float4 dyr[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int vec = li + v * LPR;
    dyr[v] = vec < d4 ? reinterpret_cast<const float4*>(dy + (int64_t)row * lddy)[vec] : f4_zero();
  }
  // pass A: d_alpha per edge (stored in de), t = sum alpha * d_alpha
  float t = 0.f;
  for (int base = start; base < end; base += G) {
    const int k = base + g;
    float p = 0.f;
    if (k < end) {
      const float4* hr = reinterpret_cast<const float4*>(h + (int64_t)col[k] * ldh);
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int vec = li + v * LPR;
        if (vec >= d4) continue;
        const float4 hv = hr[vec];
        p = fmaf(dyr[v].x, hv.x, p); p = fmaf(dyr[v].y, hv.y, p); p = fmaf(dyr[v].z, hv.z, p); p = fmaf(dyr[v].w, hv.w, p);
      }
    }
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) p += __shfl_xor(p, off);
    if (k < end && li == 0) {
      de[k] = p;
      t = fmaf(alpha[k], p, t);
    }
  }
  t = wave_sum(t);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // pass-A stores of other lanes -> pass-B loads
  __builtin_amdgcn_wave_barrier();
  // pass B
  const float ad = a_dst[row];
  float sd = 0.f;
  for (int k = start + lane; k < end; k += kWave) {
    const float s = a_src[col[k]] + ad;
    const float v = alpha[k] * (de[k] - t) * (s > 0.f ? 1.0f : slope);
    de[k] = v;
    sd += v;
  }
  sd = wave_sum(sd);
  if (lane == 0) da_dst[row] = sd;
}

// backward, source-major: dh_j = sum_{j->i} alpha_e dy_i ; da_src[j] = sum de_e
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void gat_bwd_col_kernel(const int32_t* __restrict__ rowptr_t,
                                                          const int32_t* __restrict__ col_t,
                                                          const int32_t* __restrict__ perm_t,
                                                          const float* __restrict__ alpha, const float* __restrict__ de,
                                                          const float* __restrict__ dy, int64_t lddy,
                                                          float* __restrict__ dh, int64_t lddh,
                                                          float* __restrict__ da_src, int32_t n_rows, int32_t d4) {
  constexpr int G = kWave / LPR;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int g = lane / LPR, li = lane % LPR;
  const int start = rowptr_t[row], end = rowptr_t[row + 1];
  float4 acc[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = f4_zero();
  float sa = 0.f;
  for (int base = start; base < end; base += kWave) {
    const int k = base + lane;
    const bool live = k < end;
    const int c = live ? col_t[k] : 0;
    const int p = live ? perm_t[k] : 0;
    const float w = live ? alpha[p] : 0.f;
    if (live) sa += de[p];
    const int cnt = min(kWave, end - base);
    const int trips = (cnt + G - 1) / G;

    for (int it = 0; it < trips; ++it) {
      const int j = it * G + g;
      const int cj = __shfl(c, j);
      const float wj = __shfl(w, j);
      const float4* dr = reinterpret_cast<const float4*>(dy + (int64_t)cj * lddy);
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int vec = li + v * LPR;
        if (vec < d4) acc[v] = f4_fma(wj, dr[vec], acc[v]);
      }
    }
  }
  sa = wave_sum(sa);
  if (lane == 0) da_src[row] = sa;
#pragma unroll
  for (int v = 0; v < VPL; ++v)
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) acc[v] = f4_add(acc[v], f4_shfl_xor(acc[v], off));
  if (g != 0) return;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int vec = li + v * LPR;
    if (vec < d4) reinterpret_cast<float4*>(dh + (int64_t)row * lddh)[vec] = acc[v];
  }
}

// =============================================================================================
// Load-balanced GAT (same work items as gd_spmm_csr_balanced_f32: <= 64 in-edges per piece, hub
// rows spread over many waves, XCD-contiguous item ranges).  A piece computes a LOCAL softmax
// (m_p = max score, s_p = sum exp(e - m_p), acc_p = sum exp(e - m_p) h_j); pieces of one row are
// merged flash-attention style:  M = max m_p,  S = sum s_p e^(m_p-M),  y = sum acc_p e^(m_p-M) / S.
// The row statistics (M, S) are all the backward needs to rebuild the attention weights.
// Latency structure as in spmm.hip: a wave's cost per visited item is the number of DEPENDENT round trips.
// Descriptors are fetched two visits ahead (by one lane, kept in SGPRs) and the item's column indices one visit
// ahead; the gathers of the first batch need only those indices, so they are ISSUED before the attention logits
// a_src[col] / a_dst[row] are waited for, and the local softmax (max, exp, sum) is computed while the rows are
// in flight; the bias is requested with the gathers.  One round trip per item (was: descriptor -> indices ->
// logits -> gathers -> bias).
template <int LPR, int VPL, bool EXACT, bool ADDR32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VPL == 1 ? ((ADDR32 || LPR >= 16) ? 8 : 7) : 2))) void gat_items_fwd_kernel(
    const int4* __restrict__ items, int32_t n_items, const int32_t* __restrict__ col,
    const float* __restrict__ a_src, const float* __restrict__ a_dst, const float* __restrict__ h, int64_t ldh,
    float* __restrict__ y, int64_t ldy, const float* __restrict__ bias, float* __restrict__ rowmax,
    float* __restrict__ rowsum, float* __restrict__ scratch, float* __restrict__ scratch_ms, float slope, int32_t d4,
    int32_t nnz) {
  constexpr int G = kWave / LPR;
  constexpr int U = (kWave / G) >= 4 ? 4 : (kWave / G);
  constexpr int kXcd = 8;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR, li = lane % LPR;
  const int xcd = blockIdx.x % kXcd;
  const int stride = (gridDim.x / kXcd) * 4;                        // waves per XCD
  const int wx = (blockIdx.x / kXcd) * 4 + (threadIdx.x >> 6);
  const int per = (n_items + kXcd - 1) / kXcd;
  const int i0 = xcd * per, i1 = min(n_items, i0 + per);
  int i = i0 + wx;
  if (i >= i1) return;

  uint32_t lo[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) lo[v] = 16u * (uint32_t)(EXACT ? li + v * LPR : min(li + v * LPR, d4 - 1));
  const char* bb = reinterpret_cast<const char*>(bias);
  const char* hb = reinterpret_cast<const char*>(h);
  const uint32_t pitch_b = (uint32_t)ldh * 4u;

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
  int c = col[min(d0.start + lane, nnz - 1)];
  for (; i < i1; i += stride) {
    const int row = d0.row, slot = d0.slot, cnt = d0.end - d0.start;
    const bool whole = slot < 0;
    const int c_cur = c;
    // ---- prefetch: descriptor of the visit after next, indices of the next visit
    if (lane == 0) dv = items[min(i + 2 * stride, i1 - 1)];
    c = col[min(d1.start + lane, nnz - 1)];
    // ---- this item's logits (lanes past its end never load), requested now, consumed behind the gathers
    float as = 0.f;
    if (lane < cnt) as = a_src[c_cur];
    const float ad = a_dst[row];

    float4 acc[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = f4_zero();
    const bool with_bias = bias && whole && g == 0;
    if (with_bias) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) acc[v] = *reinterpret_cast<const float4*>(bb + lo[v]);
    }
    const int trips = (cnt + G - 1) / G;
    // as in the SpMM (spmm.hip): whole trips past the end issue no gather (scalar branch); the last,
    // partly filled trip re-reads the item's last neighbour with the weight of lane `cnt` (zero)
    const int last4 = 4 * cnt - 4, end4 = 4 * cnt;
    auto gather = [&](int t, float4(&xv)[VPL]) {
      const int j4 = 4 * (t * G) + 4 * g;
      const int cs = __builtin_amdgcn_ds_bpermute(min(j4, last4), c_cur);
      if (ADDR32) {   // one full-rate 24-bit multiply, 32-bit offset on the scalar base (see spmm.hip)
        const uint32_t ro = __umul24((uint32_t)cs, pitch_b);
#pragma unroll
        for (int v = 0; v < VPL; ++v) xv[v] = *reinterpret_cast<const float4*>(hb + (ro + lo[v]));
      } else {
        const char* xr = reinterpret_cast<const char*>(h + (int64_t)cs * ldh);
#pragma unroll
        for (int v = 0; v < VPL; ++v) xv[v] = *reinterpret_cast<const float4*>(xr + lo[v]);
      }
    };
    float m = 0.f, ssum = 0.f, p_cur = 0.f;
    bool have_p = false;
    // local softmax of the item; lane group 0 starts from bias * (sum + eps): after the 1 / (sum + eps) of the
    // epilogue that is the bias
    auto softmax = [&]() {
      if (have_p) return;
      have_p = true;
      const float e = lane < cnt ? leaky(as + ad, slope) : -INFINITY;
      m = wave_max(e);
      p_cur = lane < cnt ? expf(e - m) : 0.f;
      ssum = wave_sum(p_cur);
      if (with_bias) {
        const float sden = ssum + 1e-16f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) { acc[v].x *= sden; acc[v].y *= sden; acc[v].z *= sden; acc[v].w *= sden; }
      }
    };
    auto weight = [&](int t) {
      const int j4 = 4 * (t * G) + 4 * g;
      return __int_as_float(__builtin_amdgcn_ds_bpermute(min(j4, end4), __float_as_int(p_cur)));
    };
    int t0 = 0;
    for (; t0 + U <= trips; t0 += U) {
      float4 xv[U][VPL];
#pragma unroll
      for (int u = 0; u < U; ++u) gather(t0 + u, xv[u]);
      softmax();
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float wj = weight(t0 + u);
#pragma unroll
        for (int v = 0; v < VPL; ++v) acc[v] = f4_fma(wj, xv[u][v], acc[v]);
      }
    }
    if (U > 1) {
      const int rem = trips - t0;
      if (rem > 0) {
        float4 xv[U - 1 > 0 ? U - 1 : 1][VPL];
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
          if (u < rem) gather(t0 + u, xv[u]);
        softmax();
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
          if (u < rem) {
            const float wj = weight(t0 + u);
#pragma unroll
            for (int v = 0; v < VPL; ++v) acc[v] = f4_fma(wj, xv[u][v], acc[v]);
          }
      }
    }
    softmax();                                                  // items without edges
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = f4_group_sum<LPR>(acc[v]);
    if (lane == 0) {
      if (whole) { rowmax[row] = m; rowsum[row] = ssum; }
      else { scratch_ms[2 * slot] = m; scratch_ms[2 * slot + 1] = ssum; }
    }
    if (g == 0) {
      const float inv = whole ? 1.0f / (ssum + 1e-16f) : 1.0f;
      char* ob = whole ? reinterpret_cast<char*>(y + (int64_t)row * ldy)
                       : reinterpret_cast<char*>(scratch + (int64_t)slot * d4 * 4);
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        if (!EXACT && li + v * LPR >= d4) continue;
        float4 o = acc[v];
        o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
        *reinterpret_cast<float4*>(ob + lo[v]) = o;
      }
    }
    d0 = d1;
    d1 = uniform(dv);
  }
}

__global__ __launch_bounds__(256) void gat_fixup_fwd_kernel(const int4* __restrict__ split, int32_t n_split,
                                                            const float* __restrict__ scratch,
                                                            const float* __restrict__ scratch_ms,
                                                            float* __restrict__ y, int64_t ldy,
                                                            const float* __restrict__ bias, float* __restrict__ rowmax,
                                                            float* __restrict__ rowsum, int32_t d4) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_split) return;
  const int4 sp = split[i];
  const int row = sp.x, slot0 = sp.y, n = sp.z;
  float mx, tot;
  float f_lane = 0.f;                         // n <= 64: lane s holds piece s's rescale factor e^(m_s - M)
  if (n <= kWave) {
    // the pieces' (max, sum) pairs with one load per lane instead of three dependent passes over them
    float m_s = -INFINITY, s_s = 0.f;
    if (lane < n) {
      const float2 ms = reinterpret_cast<const float2*>(scratch_ms)[slot0 + lane];
      m_s = ms.x;
      s_s = ms.y;
    }
    mx = wave_max(m_s);
    f_lane = lane < n ? expf(m_s - mx) : 0.f;
    tot = wave_sum(s_s * f_lane);
  } else {
    mx = -INFINITY;
    for (int s = 0; s < n; ++s) mx = fmaxf(mx, scratch_ms[2 * (slot0 + s)]);
    tot = 0.f;
    for (int s = 0; s < n; ++s) tot += scratch_ms[2 * (slot0 + s) + 1] * expf(scratch_ms[2 * (slot0 + s)] - mx);
  }
  const float inv = 1.0f / (tot + 1e-16f);
  for (int vec = lane; vec < d4; vec += kWave) {
    float4 o = f4_zero();
    const float4* sp4 = reinterpret_cast<const float4*>(scratch + (int64_t)slot0 * d4 * 4) + vec;
    if (n <= kWave) {
      // piece s's factor lives in lane s.  Only the lanes with vec < d4 are inside this loop, and the LDS crossbar
      // (__shfl = ds_bpermute) returns 0 for a source lane that is masked off - a hub row with more pieces than
      // d/4 lanes lost its tail (found by tests/test_full_size_gpu.py on a degree-1884 row at d = 64).  v_readlane
      // reads the register of ANY lane; s is wave-uniform.
      auto factor = [&](int s_) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f_lane), s_)); };
      int s = 0;
      for (; s + 4 <= n; s += 4) {              // four pieces in flight, accumulated in slot order
        const float4 p0 = sp4[(int64_t)s * d4], p1 = sp4[(int64_t)(s + 1) * d4], p2 = sp4[(int64_t)(s + 2) * d4],
                     p3 = sp4[(int64_t)(s + 3) * d4];
        o = f4_fma(factor(s), p0, o);
        o = f4_fma(factor(s + 1), p1, o);
        o = f4_fma(factor(s + 2), p2, o);
        o = f4_fma(factor(s + 3), p3, o);
      }
      for (; s < n; ++s) o = f4_fma(factor(s), sp4[(int64_t)s * d4], o);
    } else {
      for (int s = 0; s < n; ++s) {
        const float f = expf(scratch_ms[2 * (slot0 + s)] - mx);
        o = f4_fma(f, sp4[(int64_t)s * d4], o);
      }
    }
    o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
    if (bias) o = f4_add(o, reinterpret_cast<const float4*>(bias)[vec]);
    reinterpret_cast<float4*>(y + (int64_t)row * ldy)[vec] = o;
  }
  if (lane == 0) { rowmax[row] = mx; rowsum[row] = tot; }
}

// backward pass 1 (target-major pieces): alpha_k, d_alpha_k = <dy_i, h_j> (stored in de),
// per-piece T = sum alpha d_alpha -> t[row] (whole rows) or a scratch slot (split rows)
// Same visit structure as the forward (persistent XCD windows, descriptors two visits ahead through one lane,
// indices one visit ahead): the neighbour rows are requested from the indices alone, the attention weights
// (logits, row statistics) and the dy row arrive behind them.
template <int LPR, int VPL, bool EXACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VPL == 1 ? (LPR >= 16 ? 8 : 7) : 1, 8))) void gat_items_bwd_dalpha_kernel(
    const int4* __restrict__ items, int32_t n_items, const int32_t* __restrict__ col,
    const float* __restrict__ a_src, const float* __restrict__ a_dst, const float* __restrict__ rowmax,
    const float* __restrict__ rowsum, const float* __restrict__ h, int64_t ldh, const float* __restrict__ dy,
    int64_t lddy, float2* __restrict__ ade, float* __restrict__ t_row,
    float* __restrict__ da_dst, float* __restrict__ scratch_t, float slope, int32_t d4, int32_t nnz,
    const int32_t* __restrict__ xcd_bounds) {
  constexpr int G = kWave / LPR;
  constexpr int U = 4;                                             // neighbour rows in flight per lane group
  constexpr int kXcd = 8;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR, li = lane % LPR;
  const int xcd = blockIdx.x % kXcd;
  const int stride = (gridDim.x / kXcd) * 4;                        // waves per XCD
  const int wx = (blockIdx.x / kXcd) * 4 + (threadIdx.x >> 6);
  const int per = (n_items + kXcd - 1) / kXcd;
  const int i0 = xcd_bounds ? xcd_bounds[xcd] : xcd * per, i1 = xcd_bounds ? xcd_bounds[xcd + 1] : min(n_items, i0 + per);
  //  with a range table (= the one-launch form) hub rows are group items whose four member waves meet at block
  // barriers: only block-uniform when every limit is a multiple of 4.  A table that breaks this aborts the launch instead of
  // deadlocking it (the invariant is the caller's, include/gnndelete_hip.h)
  if (xcd_bounds && ((i0 | i1) & 3)) __builtin_trap();
  int i = i0 + wx;
  if (i >= i1) return;
  __shared__ float pbuf[4][kWave];
  __shared__ float grp_t[4], grp_v[4];
  float* pw = pbuf[threadIdx.x >> 6];
  // One-launch form (SplitPlan.onepass items, one row per item): slot == -2 = member of a GROUP (a hub row as four consecutive,
  // 4-aligned items): every member parks (alpha, <dy, h>) of its share and sums its part of t = sum alpha <dy, h>; the four
  // parts meet in LDS, and each member then finishes the score gradients of ITS edges with the row's t and sums them for
  // d a_dst - what took the pieces of a split row two more launches and two fix-up launches.  row < 0 = padding.

  uint32_t lo[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) lo[v] = 16u * (uint32_t)(EXACT ? li + v * LPR : min(li + v * LPR, d4 - 1));

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
  int c_next = col[min(d0.start + lane, nnz - 1)];
  for (; i < i1; i += stride) {
    const int row = d0.row, slot = d0.slot, rowc = max(row, 0);
    const bool member = slot == -2;
    int start = d0.start;
    int cnt = row < 0 ? 0 : min(kWave, d0.end - start);
    int c = c_next;
    if (lane == 0) dv = items[min(i + 2 * stride, i1 - 1)];
    c_next = col[min(d1.start + lane, nnz - 1)];
    // weights of the item's edges and the dy row: requested now, consumed behind the first gathers
    float as = 0.f;
    if (lane < cnt) as = a_src[c];
    const float ad = a_dst[rowc], rm = rowmax[rowc], rs = rowsum[rowc];
    float4 dyr[VPL];
    const char* dyb = reinterpret_cast<const char*>(dy + (int64_t)rowc * lddy);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      dyr[v] = *reinterpret_cast<const float4*>(dyb + lo[v]);
      if (!EXACT && li + v * LPR >= d4) dyr[v] = f4_zero();
    }
    float t_acc = 0.f;                                  // a member's part of t over its chunks
    for (;;) {                                          // 64-edge chunks: one, except for the members of rows above 256 in-edges
    const int trips = (cnt + G - 1) / G;
    const int last4 = 4 * cnt - 4;
    auto gather = [&](int t, float4(&hv)[VPL]) {
      const int j4 = min(4 * (t * G) + 4 * g, last4);              // padded slots re-read the last neighbour
      const int cj = __builtin_amdgcn_ds_bpermute(j4, c);
      const char* hr = reinterpret_cast<const char*>(h + (int64_t)cj * ldh);
#pragma unroll
      for (int v = 0; v < VPL; ++v) hv[v] = *reinterpret_cast<const float4*>(hr + lo[v]);
    };
    auto dot = [&](int t, const float4(&hv)[VPL]) {
      const int j = t * G + g;
      float p = 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        p = fmaf(dyr[v].x, hv[v].x, p); p = fmaf(dyr[v].y, hv[v].y, p);
        p = fmaf(dyr[v].z, hv[v].z, p); p = fmaf(dyr[v].w, hv[v].w, p);
      }
      p = lanes_sum<LPR>(p);
      if (j < cnt && li == 0) pw[j] = p;                            // edge j's <dy_i, h_j>, parked per edge
    };
    int t0 = 0;
    for (; t0 + U <= trips; t0 += U) {
      float4 hv[U][VPL];
#pragma unroll
      for (int u = 0; u < U; ++u) gather(t0 + u, hv[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) dot(t0 + u, hv[u]);
    }
    {
      const int rem = trips - t0;
      if (rem > 0) {
        float4 hv[U - 1][VPL];
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
          if (u < rem) gather(t0 + u, hv[u]);
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
          if (u < rem) dot(t0 + u, hv[u]);
      }
    }
    const float e = leaky(as + ad, slope);
    const float al = lane < cnt ? expf(e - rm) / (rs + 1e-16f) : 0.f;
    // one coalesced 8-byte store of the item's edge values (alpha, score gradient) - interleaved, because the
    // source-major pass gathers both through the edge permutation: one 64-byte sector per edge instead of two and the attention-weighted sum, edge per lane
    // (LDS operations of one wave complete in order: no barrier needed)
    const float pj = lane < cnt ? pw[lane] : 0.f;
    float tpart = al * pj;
    tpart = wave_sum(tpart);
    if (slot == -1) {
      // the item is the whole row: t_i = sum_j alpha_ij <dy_i, h_j> is complete, so the score gradient
      // de_ij = alpha_ij (<dy_i, h_j> - t_i) leaky'(s_ij) and da_dst_i = sum_j de_ij are finished here
      float v = lane < cnt ? al * (pj - tpart) * (as + ad > 0.f ? 1.0f : slope) : 0.f;
      if (lane < cnt) ade[start + lane] = make_float2(al, v);
      v = wave_sum(v);
      if (lane == 0 && row >= 0) { t_row[row] = tpart; da_dst[row] = v; }
      break;
    }
    if (!member) {
      // a piece of a split row: park <dy_i, h_j> and the partial t; gat_items_bwd_de_kernel finishes the
      // pieces once the row's t is summed
      if (lane < cnt) ade[start + lane] = make_float2(al, pj);
      if (lane == 0) scratch_t[slot] = tpart;
      break;
    }
    // a group member: park this chunk, go on with the next one of its share
    if (lane < cnt) ade[start + lane] = make_float2(al, pj);
    t_acc += __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(tpart)));   // (wave-uniform: a scalar register)
    start += kWave;
    if (start >= d0.end) break;
    cnt = min(kWave, d0.end - start);
    c = col[min(start + lane, nnz - 1)];
    as = lane < cnt ? a_src[c] : 0.f;
    }
    if (member) {              // (block-uniform: the planner aligns the quadruples and the XCD ranges to 4)
      const int wave = threadIdx.x >> 6;
      if (lane == 0) grp_t[wave] = t_acc;
      __syncthreads();
      const float t = ((grp_t[0] + grp_t[1]) + grp_t[2]) + grp_t[3];
      float vs = 0.f;
      for (int b = d0.start; b < d0.end; b += kWave) {           // this member's edges again: (alpha, <dy, h>) -> (alpha, de)
        const int cn = min(kWave, d0.end - b), k = min(b + lane, nnz - 1);
        float v = 0.f;
        if (lane < cn) {
          const float2 a2 = ade[k];
          const float sc = a_src[col[k]] + ad;
          v = a2.x * (a2.y - t) * (sc > 0.f ? 1.0f : slope);
          ade[k] = make_float2(a2.x, v);
        }
        vs += wave_sum(v);
      }
      if (lane == 0) grp_v[wave] = vs;
      __syncthreads();
      if (wave == 0 && lane == 0 && row >= 0) {
        t_row[row] = t;
        da_dst[row] = ((grp_v[0] + grp_v[1]) + grp_v[2]) + grp_v[3];
      }
      __syncthreads();
    }
    d0 = d1;
    d1 = uniform(dv);
  }
}

__global__ __launch_bounds__(256) void scalar_fixup_kernel(const int4* __restrict__ split, int32_t n_split,
                                                           const float* __restrict__ scratch, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_split) return;
  const int4 sp = split[i];
  float s = 0.f;
  for (int k = 0; k < sp.z; ++k) s += scratch[sp.y + k];
  out[sp.x] = s;
}

// backward pass 2 (edge-parallel inside pieces): de_k = alpha_k (d_alpha_k - t_i) leaky'(s_k); da_dst partials
__global__ __launch_bounds__(256) void gat_items_bwd_de_kernel(
    const int4* __restrict__ items, int32_t n_items, const int32_t* __restrict__ col,
    const float* __restrict__ a_src, const float* __restrict__ a_dst, float2* __restrict__ ade,
    const float* __restrict__ t_row, float* __restrict__ scratch_d, float slope, int32_t nnz) {
  // Only the pieces of split rows are left for this pass (whole rows were finished by
  // gat_items_bwd_dalpha_kernel): a wave looks at kPer consecutive descriptors and skips the whole rows.
  constexpr int kPer = 4;
  const int lane = threadIdx.x & 63;
  const int first = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kPer;
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const int item = first + q;
    if (item >= n_items) return;
    const int4 desc = items[item];
    const int row = desc.x, start = desc.y, slot = desc.w;
    if (slot < 0) continue;
    const int cnt = desc.z - start;
    float v = 0.f;
    if (lane < cnt) {
      const int k = start + lane;
      const float s = a_src[col[k]] + a_dst[row];
      const float2 ad = ade[k];
      v = ad.x * (ad.y - t_row[row]) * (s > 0.f ? 1.0f : slope);
      ade[k] = make_float2(ad.x, v);
    }
    v = wave_sum(v);
    if (lane == 0) scratch_d[slot] = v;
  }
}

// a1[i] = <h[i,:], v1>, a2[i] = <h[i,:], v2>  (the GAT attention logits a_src / a_dst): one pass over h
template <int LPR>
__global__ __launch_bounds__(256) void row_dots_kernel(const float* __restrict__ h, int64_t ldh, int32_t n,
                                                       int32_t d4, const float* __restrict__ v1,
                                                       const float* __restrict__ v2, float* __restrict__ a1,
                                                       float* __restrict__ a2) {
  constexpr int G = kWave / LPR;
  const int lane = threadIdx.x & 63;
  const int g = lane / LPR, li = lane % LPR;
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * G + g;
  float p1 = 0.f, p2 = 0.f;
  if (row < n) {
    const float4* hr = reinterpret_cast<const float4*>(h + (int64_t)row * ldh);
    for (int vec = li; vec < d4; vec += LPR) {
      const float4 hv = hr[vec];
      const float4 x1 = reinterpret_cast<const float4*>(v1)[vec];
      const float4 x2 = reinterpret_cast<const float4*>(v2)[vec];
      p1 = fmaf(hv.x, x1.x, p1); p1 = fmaf(hv.y, x1.y, p1); p1 = fmaf(hv.z, x1.z, p1); p1 = fmaf(hv.w, x1.w, p1);
      p2 = fmaf(hv.x, x2.x, p2); p2 = fmaf(hv.y, x2.y, p2); p2 = fmaf(hv.z, x2.z, p2); p2 = fmaf(hv.w, x2.w, p2);
    }
  }
  p1 = lanes_sum<LPR>(p1);
  p2 = lanes_sum<LPR>(p2);
  if (row < n && li == 0) { a1[row] = p1; a2[row] = p2; }
}

// y[i,:] += a[i] * u + b[i] * v : the part of d(lin_src output) that comes through the attention logits
// (a_src = <h, att_src>, a_dst = <h, att_dst>), both rank-1 terms in one pass over y
__global__ __launch_bounds__(256) void rank1_add2_kernel(float* __restrict__ y, int64_t ldy, int32_t n, int32_t d4,
                                                         const float* __restrict__ a, const float* __restrict__ u,
                                                         const float* __restrict__ b, const float* __restrict__ v) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)n * d4) return;
  const int row = (int)(e / d4), vec = (int)(e % d4);
  float4* yr = reinterpret_cast<float4*>(y + (int64_t)row * ldy) + vec;
  float4 o = *yr;
  o = f4_fma(a[row], reinterpret_cast<const float4*>(u)[vec], o);
  o = f4_fma(b[row], reinterpret_cast<const float4*>(v)[vec], o);
  *yr = o;
}

// out[i] = sum_{k in [rowptr[i], rowptr[i+1])} x[perm ? perm[k] : k]   (deterministic, one wave per row)
__global__ __launch_bounds__(256) void segment_sum_kernel(const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ perm,
                                                          const float* __restrict__ x, int32_t n,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  float s = 0.f;
  for (int k = rowptr[row] + lane; k < rowptr[row + 1]; k += kWave) s += x[perm ? perm[k] : k];
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

// Edge quantities of the GAT backward moved to the transposed (source-major) edge order in one pass:
//   alpha_t[k] = ade[perm[k]].x                 (the weights of the transposed SpMM that forms dh)
//   da_src[j]  = sum_{k in row j} ade[perm[k]].y (gradient of the source attention logit)
// 8 lanes per source row (8 rows per wave; the typical row has ~8 out-edges), rows with more than 256
// out-edges are redone by the whole wave.  Fixed summation order per row (deterministic).
__global__ __launch_bounds__(256) void gat_transpose_edge_kernel(const int32_t* __restrict__ rowptr_t,
                                                                 const int32_t* __restrict__ perm,
                                                                 const float2* __restrict__ ade, int32_t n,
                                                                 float* __restrict__ alpha_t,
                                                                 float* __restrict__ da_src) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 3, li = lane & 7;
  const int base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
  const int r = base + g;
  int start = 0, end = 0;
  if (r < n) { start = rowptr_t[r]; end = rowptr_t[r + 1]; }
  const bool heavy = end - start > 256;
  float s = 0.f;
  if (!heavy)
    for (int k = start + li; k < end; k += 8) {
      const float2 ad = ade[perm[k]];
      alpha_t[k] = ad.x;
      s += ad.y;
    }
  s = lanes_sum<8>(s);
  if (r < n && !heavy && li == 0) da_src[r] = s;
  unsigned long long todo = __ballot(heavy && li == 0);
  while (todo) {
    const int g2 = (__ffsll((long long)todo) - 1) >> 3;
    todo &= todo - 1;
    const int r2 = base + g2;
    const int s2 = rowptr_t[r2], e2 = rowptr_t[r2 + 1];
    float t = 0.f;
    for (int k = s2 + lane; k < e2; k += kWave) {
      const float2 ad = ade[perm[k]];
      alpha_t[k] = ad.x;
      t += ad.y;
    }
    t = wave_sum(t);
    if (lane == 0) da_src[r2] = t;
  }
}

#define GD_GAT_DISPATCH(KERNEL, ...)                                                         \
  do {                                                                                       \
    const int lpr = lanes_per_row(d4);                                                       \
    switch (lpr) {                                                                           \
      case 1: hipLaunchKernelGGL((KERNEL<1, 1>), grid, dim3(256), 0, s, __VA_ARGS__); break;  \
      case 2: hipLaunchKernelGGL((KERNEL<2, 1>), grid, dim3(256), 0, s, __VA_ARGS__); break;  \
      case 4: hipLaunchKernelGGL((KERNEL<4, 1>), grid, dim3(256), 0, s, __VA_ARGS__); break;  \
      case 8: hipLaunchKernelGGL((KERNEL<8, 1>), grid, dim3(256), 0, s, __VA_ARGS__); break;  \
      case 16: hipLaunchKernelGGL((KERNEL<16, 1>), grid, dim3(256), 0, s, __VA_ARGS__); break; \
      case 32: hipLaunchKernelGGL((KERNEL<32, 1>), grid, dim3(256), 0, s, __VA_ARGS__); break; \
      default:                                                                               \
        if (d4 <= 64) hipLaunchKernelGGL((KERNEL<64, 1>), grid, dim3(256), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<64, 4>), grid, dim3(256), 0, s, __VA_ARGS__);          \
    }                                                                                        \
  } while (0)

}  // namespace gd

extern "C" int gd_gat_aggregate_f32(const int32_t* rowptr, const int32_t* col, const float* a_src, const float* a_dst,
                                    const float* h, int64_t ldh, float* y, int64_t ldy, const float* bias,
                                    float* alpha_out, float slope, int32_t n_rows, int32_t d, void* stream) {
  using namespace gd;
  GD_REQUIRE(rowptr && col && a_src && a_dst && h && y, GD_E_NULL, "gd_gat_aggregate_f32: null pointer");
  GD_REQUIRE(n_rows >= 0 && d > 0 && d % 4 == 0 && d <= 1024 && ldh % 4 == 0 && ldy % 4 == 0, GD_E_DIM,
             "gd_gat_aggregate_f32: d=%d must be a multiple of 4 (<=1024) with 16-byte row strides", d);
  GD_REQUIRE(aligned16(h) && aligned16(y) && (!bias || aligned16(bias)), GD_E_ALIGN, "gd_gat_aggregate_f32: unaligned");
  if (n_rows == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((n_rows + 3) / 4);
  const int d4 = d / 4;
  GD_GAT_DISPATCH(gat_fwd_kernel, rowptr, col, a_src, a_dst, h, ldh, y, ldy, bias, alpha_out, slope, n_rows, d4);
  return launched("gat_fwd");
}

extern "C" int gd_gat_aggregate_bwd_f32(const int32_t* rowptr, const int32_t* col, const float* alpha,
                                        const int32_t* rowptr_t, const int32_t* col_t, const int32_t* perm_t,
                                        const float* a_src, const float* a_dst, const float* h, int64_t ldh,
                                        const float* dy, int64_t lddy, float* dh, int64_t lddh, float* da_src,
                                        float* da_dst, float* de, float slope, int32_t n_rows, int32_t d,
                                        void* stream) {
  using namespace gd;
  GD_REQUIRE(rowptr && col && alpha && rowptr_t && col_t && perm_t && a_src && a_dst && h && dy && dh && da_src &&
                 da_dst && de, GD_E_NULL, "gd_gat_aggregate_bwd_f32: null pointer");
  GD_REQUIRE(n_rows >= 0 && d > 0 && d % 4 == 0 && d <= 1024 && ldh % 4 == 0 && lddy % 4 == 0 && lddh % 4 == 0,
             GD_E_DIM, "gd_gat_aggregate_bwd_f32: bad dims");
  GD_REQUIRE(aligned16(h) && aligned16(dy) && aligned16(dh), GD_E_ALIGN, "gd_gat_aggregate_bwd_f32: unaligned");
  if (n_rows == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((n_rows + 3) / 4);
  const int d4 = d / 4;
  GD_GAT_DISPATCH(gat_bwd_row_kernel, rowptr, col, alpha, a_src, a_dst, h, ldh, dy, lddy, de, da_dst, slope, n_rows, d4);
  int rc = launched("gat_bwd_row");
  if (rc) return rc;
  GD_GAT_DISPATCH(gat_bwd_col_kernel, rowptr_t, col_t, perm_t, alpha, de, dy, lddy, dh, lddh, da_src, n_rows, d4);
  return launched("gat_bwd_col");
}

#define GD_GAT_ITEMS(KERNEL, ...)                                                                              \
  do {                                                                                                         \
    const int lpr = lanes_per_row(d4);                                                                         \
    const bool ex = (lpr == 64) ? (d4 == 64 || d4 == 256) : (d4 == lpr);                                        \
    switch (lpr) {                                                                                             \
      case 1: hipLaunchKernelGGL((KERNEL<1, 1, true>), grid, dim3(256), 0, s, __VA_ARGS__); break;             \
      case 2: hipLaunchKernelGGL((KERNEL<2, 1, true>), grid, dim3(256), 0, s, __VA_ARGS__); break;             \
      case 4: hipLaunchKernelGGL((KERNEL<4, 1, true>), grid, dim3(256), 0, s, __VA_ARGS__); break;             \
      case 8: hipLaunchKernelGGL((KERNEL<8, 1, true>), grid, dim3(256), 0, s, __VA_ARGS__); break;             \
      case 16: hipLaunchKernelGGL((KERNEL<16, 1, true>), grid, dim3(256), 0, s, __VA_ARGS__); break;           \
      case 32: hipLaunchKernelGGL((KERNEL<32, 1, true>), grid, dim3(256), 0, s, __VA_ARGS__); break;           \
      default:                                                                                                 \
        if (d4 == 64) hipLaunchKernelGGL((KERNEL<64, 1, true>), grid, dim3(256), 0, s, __VA_ARGS__);           \
        else if (d4 < 64) hipLaunchKernelGGL((KERNEL<64, 1, false>), grid, dim3(256), 0, s, __VA_ARGS__);      \
        else if (d4 == 256) hipLaunchKernelGGL((KERNEL<64, 4, true>), grid, dim3(256), 0, s, __VA_ARGS__);     \
        else hipLaunchKernelGGL((KERNEL<64, 4, false>), grid, dim3(256), 0, s, __VA_ARGS__);                   \
    }                                                                                                          \
    (void)ex;                                                                                                  \
  } while (0)
#define GD_A32_TRUE , true
#define GD_A32_FALSE , false
#define GD_GAT_ITEMS_T(KERNEL, TAIL, ...)                                                                              \
  do {                                                                                                         \
    const int lpr = lanes_per_row(d4);                                                                         \
    const bool ex = (lpr == 64) ? (d4 == 64 || d4 == 256) : (d4 == lpr);                                        \
    switch (lpr) {                                                                                             \
      case 1: hipLaunchKernelGGL((KERNEL<1, 1, true TAIL>), grid, dim3(256), 0, s, __VA_ARGS__); break;             \
      case 2: hipLaunchKernelGGL((KERNEL<2, 1, true TAIL>), grid, dim3(256), 0, s, __VA_ARGS__); break;             \
      case 4: hipLaunchKernelGGL((KERNEL<4, 1, true TAIL>), grid, dim3(256), 0, s, __VA_ARGS__); break;             \
      case 8: hipLaunchKernelGGL((KERNEL<8, 1, true TAIL>), grid, dim3(256), 0, s, __VA_ARGS__); break;             \
      case 16: hipLaunchKernelGGL((KERNEL<16, 1, true TAIL>), grid, dim3(256), 0, s, __VA_ARGS__); break;           \
      case 32: hipLaunchKernelGGL((KERNEL<32, 1, true TAIL>), grid, dim3(256), 0, s, __VA_ARGS__); break;           \
      default:                                                                                                 \
        if (d4 == 64) hipLaunchKernelGGL((KERNEL<64, 1, true TAIL>), grid, dim3(256), 0, s, __VA_ARGS__);           \
        else if (d4 < 64) hipLaunchKernelGGL((KERNEL<64, 1, false TAIL>), grid, dim3(256), 0, s, __VA_ARGS__);      \
        else if (d4 == 256) hipLaunchKernelGGL((KERNEL<64, 4, true TAIL>), grid, dim3(256), 0, s, __VA_ARGS__);     \
        else hipLaunchKernelGGL((KERNEL<64, 4, false TAIL>), grid, dim3(256), 0, s, __VA_ARGS__);                   \
    }                                                                                                          \
    (void)ex;                                                                                                  \
  } while (0)

extern "C" int64_t gd_gat_balanced_scratch(int32_t n_slots, int32_t d) { return (int64_t)n_slots * (d + 4) + 4; }

extern "C" int gd_gat_aggregate_balanced_f32(const int32_t* items, int32_t n_items, const int32_t* split,
                                             int32_t n_split, int32_t n_slots, const int32_t* col, const float* a_src,
                                             const float* a_dst, const float* h, int64_t ldh, float* y, int64_t ldy,
                                             const float* bias, float* rowmax, float* rowsum, float* scratch,
                                             float slope, int32_t d, int32_t nnz, int32_t h_rows, void* stream) {
  using namespace gd;
  GD_REQUIRE(items && col && a_src && a_dst && h && y && rowmax && rowsum, GD_E_NULL,
             "gd_gat_aggregate_balanced_f32: null pointer");
  GD_REQUIRE(n_split == 0 || (split && scratch), GD_E_NULL, "gd_gat_aggregate_balanced_f32: split rows need scratch");
  GD_REQUIRE(d > 0 && d % 4 == 0 && d <= 1024 && ldh % 4 == 0 && ldy % 4 == 0 && (d & (d - 1)) == 0, GD_E_DIM,
             "gd_gat_aggregate_balanced_f32: d=%d must be a power of two in [4,1024]", d);
  GD_REQUIRE(aligned16(h) && aligned16(y) && aligned16(items) && (!bias || aligned16(bias)) &&
                 (!scratch || aligned16(scratch)), GD_E_ALIGN, "gd_gat_aggregate_balanced_f32: unaligned pointer");
  if (n_items == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const int d4 = d / 4;
  int nblk = (n_items + 3) / 4;
  if (nblk > 8192) nblk = 8192;
  nblk = (nblk + 7) / 8 * 8;
  const dim3 grid(nblk);
  float* scratch_ms = scratch ? scratch + (int64_t)n_slots * d : nullptr;
  const int4* it = reinterpret_cast<const int4*>(items);
  // 24-bit fast addressing (see spmm.hip): row ids and the row pitch in bytes below 2^24, h smaller than 4 GiB
  const bool addr32 = h_rows > 0 && h_rows <= (1 << 24) && ldh * 4 < (1 << 24) && (int64_t)h_rows * ldh * 4 < (1ll << 32);
  if (addr32)
    GD_GAT_ITEMS_T(gat_items_fwd_kernel, GD_A32_TRUE, it, n_items, col, a_src, a_dst, h, ldh, y, ldy, bias, rowmax, rowsum,
                   scratch, scratch_ms, slope, d4, nnz);
  else
    GD_GAT_ITEMS_T(gat_items_fwd_kernel, GD_A32_FALSE, it, n_items, col, a_src, a_dst, h, ldh, y, ldy, bias, rowmax, rowsum,
                   scratch, scratch_ms, slope, d4, nnz);
  int rc = launched("gat_items_fwd");
  if (rc || n_split == 0) return rc;
  hipLaunchKernelGGL(gat_fixup_fwd_kernel, dim3((n_split + 3) / 4), dim3(256), 0, s,
                     reinterpret_cast<const int4*>(split), n_split, scratch, scratch_ms, y, ldy, bias, rowmax, rowsum, d4);
  return launched("gat_fixup_fwd");
}

extern "C" int gd_gat_edge_grads_balanced_f32(const int32_t* items, int32_t n_items, const int32_t* split,
                                              int32_t n_split, const int32_t* col, const float* a_src,
                                              const float* a_dst, const float* rowmax, const float* rowsum,
                                              const float* h, int64_t ldh, const float* dy, int64_t lddy,
                                              float* ade, float* da_dst, float* t_row, float* scratch,
                                              float slope, int32_t d, int32_t nnz, const int32_t* xcd_bounds, void* stream) {
  using namespace gd;
  GD_REQUIRE(items && col && a_src && a_dst && rowmax && rowsum && h && dy && ade && da_dst && t_row, GD_E_NULL,
             "gd_gat_edge_grads_balanced_f32: null pointer");
  GD_REQUIRE(!xcd_bounds || (n_split == 0 && n_items % 4 == 0), GD_E_DIM,
             "gd_gat_edge_grads_balanced_f32: xcd_bounds go with one-launch items (groups, n_items a multiple of 4) and no split list");
  GD_REQUIRE(n_split == 0 || (split && scratch), GD_E_NULL, "gd_gat_edge_grads_balanced_f32: split rows need scratch");
  GD_REQUIRE(d > 0 && d % 4 == 0 && d <= 1024 && ldh % 4 == 0 && lddy % 4 == 0 && (d & (d - 1)) == 0, GD_E_DIM,
             "gd_gat_edge_grads_balanced_f32: d=%d must be a power of two in [4,1024]", d);
  GD_REQUIRE(aligned16(h) && aligned16(dy) && aligned16(items) && (reinterpret_cast<uintptr_t>(ade) & 7u) == 0, GD_E_ALIGN,
             "gd_gat_edge_grads_balanced_f32: unaligned");
  if (n_items == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const int d4 = d / 4;
  int nblk = (n_items + 3) / 4;
  if (nblk > 8192) nblk = 8192;                         // persistent visits, as the forward
  const dim3 grid((nblk + 7) / 8 * 8);
  const int4* it = reinterpret_cast<const int4*>(items);
  const int4* sp = reinterpret_cast<const int4*>(split);
  GD_GAT_ITEMS(gat_items_bwd_dalpha_kernel, it, n_items, col, a_src, a_dst, rowmax, rowsum, h, ldh, dy, lddy,
               reinterpret_cast<float2*>(ade), t_row, da_dst, scratch, slope, d4, nnz, xcd_bounds);
  int rc = launched("gat_items_bwd_dalpha");
  if (rc) return rc;
  if (n_split) {
    hipLaunchKernelGGL(scalar_fixup_kernel, dim3((n_split + 255) / 256), dim3(256), 0, s, sp, n_split, scratch, t_row);
    if ((rc = launched("gat_fixup_t"))) return rc;
  }
  if (n_split) {
    // only the pieces of split rows are left (whole rows exit at once)
    hipLaunchKernelGGL(gat_items_bwd_de_kernel, dim3((n_items + 15) / 16), dim3(256), 0, s, it, n_items, col, a_src,
                       a_dst, reinterpret_cast<float2*>(ade), t_row, scratch, slope, nnz);
    if ((rc = launched("gat_items_bwd_de"))) return rc;
    hipLaunchKernelGGL(scalar_fixup_kernel, dim3((n_split + 255) / 256), dim3(256), 0, s, sp, n_split, scratch, da_dst);
    rc = launched("gat_fixup_dadst");
  }
  return rc;
}

extern "C" int gd_row_dots_f32(const float* h, int64_t ldh, int32_t n, int32_t d, const float* v1, const float* v2,
                               float* a1, float* a2, void* stream) {
  using namespace gd;
  GD_REQUIRE(h && v1 && v2 && a1 && a2, GD_E_NULL, "gd_row_dots_f32: null pointer");
  GD_REQUIRE(n >= 0 && d > 0 && d % 4 == 0 && ldh % 4 == 0, GD_E_DIM, "gd_row_dots_f32: d must be a multiple of 4");
  GD_REQUIRE(aligned16(h) && aligned16(v1) && aligned16(v2), GD_E_ALIGN, "gd_row_dots_f32: unaligned pointer");
  if (n == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const int d4 = d / 4;
  const int lpr = lanes_per_row(d4);
  const int per_block = 4 * (kWave / lpr);
  const dim3 grid((n + per_block - 1) / per_block);
  switch (lpr) {
    case 1: hipLaunchKernelGGL((row_dots_kernel<1>), grid, dim3(256), 0, s, h, ldh, n, d4, v1, v2, a1, a2); break;
    case 2: hipLaunchKernelGGL((row_dots_kernel<2>), grid, dim3(256), 0, s, h, ldh, n, d4, v1, v2, a1, a2); break;
    case 4: hipLaunchKernelGGL((row_dots_kernel<4>), grid, dim3(256), 0, s, h, ldh, n, d4, v1, v2, a1, a2); break;
    case 8: hipLaunchKernelGGL((row_dots_kernel<8>), grid, dim3(256), 0, s, h, ldh, n, d4, v1, v2, a1, a2); break;
    case 16: hipLaunchKernelGGL((row_dots_kernel<16>), grid, dim3(256), 0, s, h, ldh, n, d4, v1, v2, a1, a2); break;
    case 32: hipLaunchKernelGGL((row_dots_kernel<32>), grid, dim3(256), 0, s, h, ldh, n, d4, v1, v2, a1, a2); break;
    default: hipLaunchKernelGGL((row_dots_kernel<64>), grid, dim3(256), 0, s, h, ldh, n, d4, v1, v2, a1, a2); break;
  }
  return launched("row_dots");
}

extern "C" int gd_segment_sum_f32(const int32_t* rowptr, const int32_t* perm, const float* x, int32_t n, float* out,
                                  void* stream) {
  using namespace gd;
  GD_REQUIRE(rowptr && x && out, GD_E_NULL, "gd_segment_sum_f32: null pointer");
  if (n <= 0) return GD_OK;
  hipLaunchKernelGGL(segment_sum_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, rowptr, perm, x, n, out);
  return launched("segment_sum");
}

extern "C" int gd_rank1_add2_f32(float* y, int64_t ldy, int32_t n, int32_t d, const float* a, const float* u,
                                 const float* b, const float* v, void* stream) {
  using namespace gd;
  GD_REQUIRE(y && a && u && b && v, GD_E_NULL, "gd_rank1_add2_f32: null pointer");
  GD_REQUIRE(n >= 0 && d > 0 && d % 4 == 0 && ldy % 4 == 0 && ldy >= d, GD_E_DIM, "gd_rank1_add2_f32: d must be a multiple of 4");
  GD_REQUIRE(aligned16(y) && aligned16(u) && aligned16(v), GD_E_ALIGN, "gd_rank1_add2_f32: unaligned pointer");
  if (n == 0) return GD_OK;
  const int64_t total = (int64_t)n * (d / 4);
  hipLaunchKernelGGL(rank1_add2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, ldy,
                     n, d / 4, a, u, b, v);
  return launched("rank1_add2");
}

extern "C" int gd_gat_transpose_edges_f32(const int32_t* rowptr_t, const int32_t* perm, const float* ade, int32_t n,
                                          float* alpha_t, float* da_src, void* stream) {
  using namespace gd;
  if (n <= 0) return GD_OK;
  GD_REQUIRE(rowptr_t && perm && ade && alpha_t && da_src && (reinterpret_cast<uintptr_t>(ade) & 7u) == 0, GD_E_NULL,
             "gd_gat_transpose_edges_f32: null or unaligned pointer");
  hipLaunchKernelGGL(gat_transpose_edge_kernel, dim3((n + 31) / 32), dim3(256), 0, (hipStream_t)stream, rowptr_t, perm,
                     reinterpret_cast<const float2*>(ade), n, alpha_t, da_src);
  return launched("gat_transpose_edges");
}
