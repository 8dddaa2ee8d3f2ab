// Weight-stationary row GEMM for the Del operator's widths (d_in, d_out in {64, 128}):
//
//     out[row(s), :] = act(in[row(s), :]) @ W        s over an index list (or all rows)
//
// The LDS-operand form (rows_gemm.hip: four waves per SIMD, the weight operand of every matrix instruction read from a 64 KB
// LDS image) holds the fp32 matrix pipe 50-57 % busy; its knock-outs put a quarter of the time on that LDS operand (NOTES,
// rounds 1-3).  Here ONE wave per SIMD keeps the WHOLE weight as MFMA A-fragments in its registers (128 x 128 = 256 of the 512
// registers a lone wave owns) and nothing but matrix instructions sits between a row tile's loads and its stores:
//
//   v_mfma_f32_16x16x4_f32, work unit = 16 rows.  lane (r = lane & 15, kq = lane >> 4) holds x[row r][kq KQ + s], s = 0 .. KQ-1
//   (KQ = d_in / 4: contiguous -> 16-byte loads); step s multiplies A = W[kq KQ + s][16 t + r] (register wr[t][s]) with it - the
//   k order is permuted, both operands agree; D: lane (r, kq) ends with outputs 16 t + 4 kq + c of ITS row: float4 stores, a
//   store instruction writes 64 contiguous bytes of 16 rows.
//   Ring: as soon as the matrix instructions of a k chunk are issued its row registers are reloaded with the same chunk of the
//   NEXT unit (a whole unit, ~3.5 us, to land); row ids / selectors are requested two units ahead.  Two accumulator sets take
//   alternate units: a unit's results are stored one output tile per k chunk UNDER the next unit's matrix instructions (a
//   lone wave that stalls on a store queue idles its matrix pipe: stores at the end of a tile cost 10 %).
//   No branch in the loop: a load or store under a branch makes every wait of the loop a vmcnt(0) (measured in the lab:
//   tools/experiments/ws_gemm_lab.hip).  Rows past the end are clamped to the last row: those lanes recompute it and rewrite
//   it with identical values.  On its first trip a wave stores zeros to its first unit's rows, overwritten a trip later by the
//   results (same wave, same addresses: program order).
//   The weight reaches the registers through LDS once per block (coalesced pass over W, then conflict-free 4-byte reads).
//
// Measured (lab, 178,921 x 128 -> 128, back to back): 59.7 us against 75.8 us for the LDS-operand form; 235,868 x 128 -> 128:
// 73.6 against 95.8 us; 128 -> 64: 39.8 against 51.0 us.  Steady state 9.0-9.1 k cycles per unit (8,192 = matrix pipe only).
#include <stdlib.h>

#include "common.h"

namespace gd {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// v[l] | v[l ^ 16] | v[l ^ 32] | v[l ^ 48] in every lane (the four lanes of a row), VALU only
__device__ __forceinline__ uint32_t or_row_lanes(uint32_t v) {
  const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = a[0] | a[1];
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return b[0] | b[1];
}

// MODE 0: plain, 1: also emit the packed [out > 0] pattern of each row, 2: gate the output by such a pattern,
//      3: plain + the two row dot products o1[row] = <out[row], u1>, o2[row] = <out[row], u2> (GATConv's attention logits,
//         gd_rows_gemm_dots_f32), 4: gated like 2 after the rank-1 correction out[row, n] += o1[row] u1[n] + o2[row] u2[n]
//         (gd_rows_gemm_gated_rank1_f32: o1 / o2 are per-row scalars READ by row id), 5: plain + bias (u1 = the bias vector: the
//         accumulators of a unit start from it instead of from zero), 6: ACCUMULATE (round 6, gd_rows_gemm_accumulate_f32):
//         out[row] += x[row] @ W - the accumulators of a unit start from the rows of `out` themselves, fetched a unit ahead into the
//         registers the previous unit's first matrix instructions have just read (GraphSAGE's root term x W_r^T added onto the
//         aggregated neighbour term here, in a matrix-bound kernel with memory to spare, instead of as a second row stream of
//         the latency-bound aggregation)
// SEL: row r comes from in_alt where sel[r] != 0.  With an index list as well (round 6: the rows-only step's t2 / GAT logits
//      products) the selector byte of a listed row is a DEPENDENT load behind its row id: the ids are requested three units
//      ahead, the selector two units ahead from the id that has landed by then, and the pair travels as one descriptor word
//      (row id | selector << 31) - no wait on a fresh load anywhere in the loop.
// `out` is a restrict pointer in every mode that only writes it (what the tuned instantiations were compiled with); the
// accumulate mode reads the rows it later overwrites and declares it plain
template <bool READS_OUT> struct WsOutPtr { typedef float* __restrict__ type; };
template <> struct WsOutPtr<true> { typedef float* type; };

template <int DIN, int DOUT, int MODE, bool HASIDX, bool SEL, bool RELU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rows_gemm_ws_kernel(
    const float* __restrict__ in, int64_t ld_in, const int32_t* __restrict__ idx, int32_t n_sel, const float* __restrict__ w,
    int32_t trans_w, typename WsOutPtr<MODE == 6>::type out, int64_t ld_out, const uint32_t* __restrict__ gate_bits,
    uint32_t* __restrict__ sign_out, const float* __restrict__ in_alt, const uint8_t* __restrict__ sel,
    const float* __restrict__ u1, const float* __restrict__ u2, float* o1, float* o2) {
  constexpr int KQ = DIN / 4, NT = DOUT / 16, XV = KQ / 4, NW = DOUT / 32;
  constexpr bool GATE = MODE == 2 || MODE == 4, DOTS = MODE == 3, RANK1 = MODE == 4, BIAS = MODE == 5, ACC = MODE == 6;
  constexpr bool BOTH = HASIDX && SEL;
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (scalar loop control)
  const int r = lane & 15, kq = lane >> 4;
  const int n_units = (n_sel + 15) >> 4;
  const int n_waves = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
  // contiguous unit range per wave (the index lists are sorted: a wave's rows are neighbours)
  const int u_lo = (int)((int64_t)n_units * wid / n_waves), u_hi = (int)((int64_t)n_units * (wid + 1) / n_waves);
  const int u_first = min(u_lo, n_units - 1);

  // what a unit's row loads need and cannot compute: the row id (index list) or the row's selector byte
  auto slot_of = [&](int u) -> int { return min(u * 16 + r, n_sel - 1); };
  auto id_of = [&](int u) -> int32_t { return idx[slot_of(min(u, n_units - 1))]; };
  auto with_sel = [&](int32_t row) -> int32_t { return row | (sel[row] ? (int32_t)0x80000000 : 0); };
  auto desc_of = [&](int u) -> int32_t {
    const int s_ = slot_of(min(u, n_units - 1));
    if (BOTH) return with_sel(idx[s_]);
    return HASIDX ? idx[s_] : (SEL ? (int32_t)sel[s_] : 0);
  };
  auto row_of = [&](int32_t desc) -> int32_t { return BOTH ? (desc & 0x7fffffff) : desc; };
  auto src_of = [&](int u, int32_t desc) -> const float4* {
    const int64_t row = HASIDX ? (int64_t)row_of(desc) : (int64_t)slot_of(min(u, n_units - 1));
    const float* base = (SEL && (BOTH ? desc < 0 : desc != 0)) ? in_alt : in;
    return reinterpret_cast<const float4*>(base + row * ld_in + kq * KQ);
  };
  int32_t d_cur = desc_of(u_first), d_nxt = desc_of(u_first + 1);
  int32_t id_nn = BOTH ? id_of(u_first + 2) : 0;             // (BOTH) row id of the unit after next, its selector still to fetch
  __builtin_amdgcn_sched_barrier(0);

  // ---- weight image wl[k DOUT + n + 16 (k / KQ)]: the two k quarters one 32-lane LDS access touches sit 16 banks apart
  constexpr int FV = DIN * DOUT / 4 / 256;                 // float4 per thread
  float4 fill[FV];
  if (!trans_w) {
#pragma unroll
    for (int q = 0; q < FV; ++q) fill[q] = reinterpret_cast<const float4*>(w)[tid + 256 * q];
  } else {                                                 // [n][k] weights: thread -> (k4, n), 16 bytes along k
#pragma unroll
    for (int q = 0; q < FV; ++q) {
      const int e = tid + 256 * q, n = e % DOUT, k4 = e / DOUT;
      fill[q] = *reinterpret_cast<const float4*>(w + (int64_t)n * DIN + 4 * k4);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // the first unit's rows: in flight under the weight fill
  float4 x[XV];
  {
    const float4* src = src_of(u_first, d_cur);
#pragma unroll
    for (int i = 0; i < XV; ++i) x[i] = src[i];
  }
  __builtin_amdgcn_sched_barrier(0);
  if (!trans_w) {
#pragma unroll
    for (int q = 0; q < FV; ++q) {
      const int e = tid + 256 * q, k = e / (DOUT / 4), n4 = e % (DOUT / 4);
      *reinterpret_cast<float4*>(wl + k * DOUT + 4 * n4 + 16 * (k / KQ)) = fill[q];
    }
  } else {
#pragma unroll
    for (int q = 0; q < FV; ++q) {
      const int e = tid + 256 * q, n = e % DOUT, k4 = e / DOUT;
      const float v[4] = {fill[q].x, fill[q].y, fill[q].z, fill[q].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) wl[(4 * k4 + c) * DOUT + n + 16 * ((4 * k4 + c) / KQ)] = v[c];
    }
  }
  __syncthreads();
  float wr[NT][KQ];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < KQ; ++s) wr[t][s] = wl[(kq * KQ + s) * DOUT + 16 * t + r + 16 * kq];
  if (u_lo >= u_hi) return;

  // Two accumulator sets take alternate units: while one receives a unit's products the other one's - the previous unit's
  // results - are stored, one output tile per k chunk, under the matrix instructions (no copies; the loop body is two units,
  // an odd unit is peeled off in front).
  f32x4v acc_a[NT], acc_b[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc_a[t] = acc_b[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
  float* dst_prev = out + (HASIDX ? (int64_t)row_of(d_cur) : (int64_t)slot_of(u_lo)) * ld_out + 4 * kq;
  int sa_prev = slot_of(u_lo);
  int row_prev = HASIDX ? row_of(d_cur) : slot_of(u_lo);    // row id of the unit whose results are stored next (MODE 3 / 4)
  uint32_t sg[NW], gsh[NW];
#pragma unroll
  for (int q = 0; q < NW; ++q) sg[q] = gsh[q] = 0;
  // MODE 3 / 4: u1 / u2 at this lane's output columns 16 t + 4 kq + c; MODE 3: the partial dots of the unit being stored;
  // MODE 4: the row scalars of the unit being stored
  float4 uv1[(DOTS || RANK1) ? NT : 1], uv2[(DOTS || RANK1) ? NT : 1];
  float dp1 = 0.f, dp2 = 0.f, ra_prev = 0.f, rb_prev = 0.f;
  f32x4v ini[ACC ? NT : 1];                                 // MODE 6: the out rows of the unit about to be multiplied
  if (ACC) {
    const float* ip = out + (HASIDX ? (int64_t)row_of(d_cur) : (int64_t)slot_of(u_first)) * ld_out + 4 * kq;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 v4 = *reinterpret_cast<const float4*>(ip + 16 * t);
      ini[t] = f32x4v{v4.x, v4.y, v4.z, v4.w};
    }
  }
  if (ACC) {                                                // (the first trip's store of the idle set then rewrites the rows as they are)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc_a[t] = acc_b[t] = ini[t];
  }
  f32x4v bz[BIAS ? NT : 1];                                 // MODE 5: the bias at this lane's output columns
  if (BIAS) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 b4 = *reinterpret_cast<const float4*>(u1 + 16 * t + 4 * kq);
      bz[t] = f32x4v{b4.x, b4.y, b4.z, b4.w};
    }
  }
  if (DOTS || RANK1) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      uv1[t] = *reinterpret_cast<const float4*>(u1 + 16 * t + 4 * kq);
      uv2[t] = *reinterpret_cast<const float4*>(u2 + 16 * t + 4 * kq);
    }
  }
  if (RANK1) {
    ra_prev = o1[row_prev];
    rb_prev = o2[row_prev];
  }

  auto store_tile = [&](const f32x4v& a, int t) {
    float4 v = make_float4(a[0], a[1], a[2], a[3]);
    if (RANK1) {
      v.x = fmaf(ra_prev, uv1[t].x, fmaf(rb_prev, uv2[t].x, v.x)); v.y = fmaf(ra_prev, uv1[t].y, fmaf(rb_prev, uv2[t].y, v.y));
      v.z = fmaf(ra_prev, uv1[t].z, fmaf(rb_prev, uv2[t].z, v.z)); v.w = fmaf(ra_prev, uv1[t].w, fmaf(rb_prev, uv2[t].w, v.w));
    }
    // bit b of word q of a row's packed pattern is output 32 q + b: this lane owns bits 16 (t & 1) + 4 kq + c of word t >> 1
    if (GATE) {
      const uint32_t m = gsh[t >> 1] >> (16 * (t & 1) + 4 * kq);
      v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f;
      v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
    }
    if (MODE == 1) {
      const uint32_t b = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
      sg[t >> 1] |= b << (16 * (t & 1) + 4 * kq);
    }
    if (DOTS) {
      dp1 = fmaf(v.x, uv1[t].x, fmaf(v.y, uv1[t].y, fmaf(v.z, uv1[t].z, fmaf(v.w, uv1[t].w, dp1))));
      dp2 = fmaf(v.x, uv2[t].x, fmaf(v.y, uv2[t].y, fmaf(v.z, uv2[t].z, fmaf(v.w, uv2[t].w, dp2))));
    }
    *reinterpret_cast<float4*>(dst_prev + 16 * t) = v;
  };
  // MODE 3: the four lanes of a row add their partial dots (VALU permlane swaps); every lane stores - lane kq the value
  // kq & 1 - so that no store sits under a branch (kq 2 / 3 rewrite what kq 0 / 1 wrote)
  auto store_dots = [&]() {
    const float s1 = xor32_sum(xor16_sum(dp1)), s2 = xor32_sum(xor16_sum(dp2));
    float* dst = (kq & 1) ? o2 : o1;
    dst[row_prev] = (kq & 1) ? s2 : s1;
    dp1 = dp2 = 0.f;
  };
  auto store_signs = [&]() {          // the four lanes of a row merge their bits; lane kq writes word kq (mod NW)
    uint32_t mine = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const uint32_t v = or_row_lanes(sg[q]);
      if ((kq & (NW - 1)) == q) mine = v;
      sg[q] = 0;
    }
    sign_out[(int64_t)sa_prev * NW + (kq & (NW - 1))] = mine;
  };
  // one unit: products into acc, the previous unit's results (old) out, the next unit's rows in
  auto unit = [&](int u, f32x4v (&acc)[NT], const f32x4v (&old)[NT]) {
    int32_t d_nn, id_n3 = 0;
    if (BOTH) {
      id_n3 = id_of(u + 3);                                  // lands during this unit
      d_nn = with_sel(id_nn);                                // the id requested a unit ago: no wait
    } else {
      d_nn = desc_of(u + 2);                                 // used a unit from now
    }
    const float4* nsrc = src_of(u + 1, d_nxt);
    uint32_t gcur[NW];
    if (GATE) {
      const uint32_t* gp = gate_bits + (int64_t)slot_of(u) * NW;
#pragma unroll
      for (int q = 0; q < NW; ++q) gcur[q] = gp[q];
    }
    const int row_cur = HASIDX ? row_of(d_cur) : slot_of(u);
    float ra_cur = 0.f, rb_cur = 0.f;
    if (RANK1) {
      ra_cur = o1[row_cur];
      rb_cur = o2[row_cur];
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      float xv[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
      if (RELU) {
#pragma unroll
        for (int c = 0; c < 4; ++c) xv[c] = fmaxf(xv[c], 0.f);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (i == 0 && c == 0) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t][0], xv[0], ACC ? ini[t] : (BIAS ? bz[t] : f32x4v{0.f, 0.f, 0.f, 0.f}), 0, 0, 0);
          else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t][4 * i + c], xv[c], acc[t], 0, 0, 0);
        }
        if (c == 1) {                                        // mid-chunk: the stores of this slot
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < NT; ++t)
            if (t * XV / NT == i) store_tile(old[t], t);
          if (MODE == 1 && i == XV - 1) store_signs();
          if (DOTS && i == XV - 1) store_dots();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      x[i] = nsrc[i];                                        // the registers just consumed: same chunk of the next unit
      if (ACC && i == 0) {                                   // ... and the next unit's out rows (its accumulators' start)
        const float* ip = out + (HASIDX ? (int64_t)row_of(d_nxt) : (int64_t)slot_of(min(u + 1, n_units - 1))) * ld_out + 4 * kq;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 v4 = *reinterpret_cast<const float4*>(ip + 16 * t);
          ini[t] = f32x4v{v4.x, v4.y, v4.z, v4.w};
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (GATE) {
#pragma unroll
      for (int q = 0; q < NW; ++q) gsh[q] = gcur[q];
    }
    if (RANK1) {
      ra_prev = ra_cur;
      rb_prev = rb_cur;
    }
    row_prev = row_cur;
    dst_prev = out + (HASIDX ? (int64_t)row_of(d_cur) : (int64_t)slot_of(u)) * ld_out + 4 * kq;
    sa_prev = slot_of(u);
    d_cur = d_nxt;
    d_nxt = d_nn;
    if (BOTH) id_nn = id_n3;
  };
  int u = u_lo;
  if ((u_hi - u_lo) & 1) {
    unit(u, acc_a, acc_b);
    ++u;
  }
  for (; u < u_hi; u += 2) {
    unit(u, acc_b, acc_a);
    unit(u + 1, acc_a, acc_b);
  }
  // the last unit's results are in acc_a either way
#pragma unroll
  for (int t = 0; t < NT; ++t) store_tile(acc_a[t], t);
  if (MODE == 1) store_signs();
  if (DOTS) store_dots();
}

int ws_cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  return n;
}

struct WsEpi { const float* u1; const float* u2; float* o1; float* o2; };     // MODE 3 / 4 operands (RowDots of rows_gemm.hip)

template <int DIN, int DOUT, int MODE, bool HASIDX, bool SEL, bool RELU>
static int ws_launch(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel, const float* w, int32_t trans_w, float* out,
                     int64_t ld_out, const uint32_t* gate_bits, uint32_t* sign_out, const float* in_alt, const uint8_t* sel,
                     WsEpi epi, hipStream_t s) {
  auto kern = rows_gemm_ws_kernel<DIN, DOUT, MODE, HASIDX, SEL, RELU>;
  constexpr size_t lds = (size_t)(DIN * DOUT + 64) * sizeof(float);
  static const hipError_t once = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  GD_REQUIRE(once == hipSuccess, -(int)once, "gd_rows_gemm_f32: cannot raise the LDS limit of the weight-stationary kernel");
  hipLaunchKernelGGL(kern, dim3(ws_cu_count()), dim3(256), lds, s, in, ld_in, idx, n_sel, w, trans_w, out, ld_out, gate_bits,
                     sign_out, in_alt, sel, epi.u1, epi.u2, epi.o1, epi.o2);
  return launched("rows_gemm_ws");
}

template <int DIN, int DOUT>
static int ws_dispatch(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel, const float* w, int32_t trans_w,
                       int32_t relu_in, float* out, int64_t ld_out, const uint32_t* gate_bits, uint32_t* sign_out,
                       const float* in_alt, const uint8_t* sel, WsEpi epi, const float* bias, hipStream_t s) {
#define GD_WS(MODE, HASIDX, SELV, RELU) \
  return ws_launch<DIN, DOUT, MODE, HASIDX, SELV, RELU>(in, ld_in, idx, n_sel, w, trans_w, out, ld_out, gate_bits, sign_out, in_alt, sel, epi, s)
  if (epi.u1) {
    // the epilogue forms exist where the GAT step uses them: row dots behind a 128-wide input, the rank-1 + gate form in front
    // of a 128-wide output (everything else: the LDS-operand kernel)
    if (gate_bits) {
      if constexpr (DOUT == 128) { if (idx) GD_WS(4, true, false, false); else GD_WS(4, false, false, false); }
      return 1;
    }
    if constexpr (DIN == 128) {
      if (sel && idx) { if (relu_in) GD_WS(3, true, true, true); else GD_WS(3, true, true, false); }
      if (sel) { if (relu_in) GD_WS(3, false, true, true); else GD_WS(3, false, true, false); }
      if (idx) { if (relu_in) GD_WS(3, true, false, true); else GD_WS(3, true, false, false); }
      if (relu_in) GD_WS(3, false, false, true);
      GD_WS(3, false, false, false);
    }
    return 1;
  }
  if (gate_bits) { if (idx) GD_WS(2, true, false, false); else GD_WS(2, false, false, false); }
  if (sign_out) { if (idx) GD_WS(1, true, false, false); else GD_WS(1, false, false, false); }
  if (bias) {                                              // (dense or index list; the selector form has no biased caller)
    epi.u1 = bias;
    if (sel) return 1;
    if (idx) { if (relu_in) GD_WS(5, true, false, true); else GD_WS(5, true, false, false); }
    if (relu_in) GD_WS(5, false, false, true);
    GD_WS(5, false, false, false);
  }
  if (sel && idx) { if (relu_in) GD_WS(0, true, true, true); else GD_WS(0, true, true, false); }
  if (sel) { if (relu_in) GD_WS(0, false, true, true); else GD_WS(0, false, true, false); }
  if (idx) { if (relu_in) GD_WS(0, true, false, true); else GD_WS(0, true, false, false); }
  if (relu_in) GD_WS(0, false, false, true);
  GD_WS(0, false, false, false);
#undef GD_WS
}

static bool ws_on();
static int ws_min_rows();

// out[row(s), :] += in[row(s), :] @ W on the weight-stationary form (MODE 6); -> 1 when the shape is not covered
int rows_gemm_ws_accumulate(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel, const float* w, int32_t d_in,
                            int32_t d_out, int32_t trans_w, float* out, int64_t ld_out, void* stream) {
  if (!ws_on() || matrix_split() != 0 || n_sel < ws_min_rows()) return 1;
  if (!((d_in == 64 || d_in == 128) && (d_out == 64 || d_out == 128))) return 1;
  if (in == out || !aligned16(in) || !aligned16(out) || !aligned16(w) || ld_in % 4 || ld_out % 4) return 1;
  hipStream_t s = (hipStream_t)stream;
  const WsEpi epi{nullptr, nullptr, nullptr, nullptr};
#define GD_WS_ACC(DI, DO)                                                                                                   \
  do {                                                                                                                      \
    if (idx) return ws_launch<DI, DO, 6, true, false, false>(in, ld_in, idx, n_sel, w, trans_w, out, ld_out, nullptr, nullptr, nullptr, nullptr, epi, s); \
    return ws_launch<DI, DO, 6, false, false, false>(in, ld_in, idx, n_sel, w, trans_w, out, ld_out, nullptr, nullptr, nullptr, nullptr, epi, s);         \
  } while (0)
  if (d_in == 128 && d_out == 128) GD_WS_ACC(128, 128);
  if (d_in == 128 && d_out == 64) GD_WS_ACC(128, 64);
  if (d_in == 64 && d_out == 128) GD_WS_ACC(64, 128);
  GD_WS_ACC(64, 64);
#undef GD_WS_ACC
}

static bool ws_on() {
  static const bool on = [] { const char* e = getenv("GD_ROWS_GEMM_WS"); return !(e && atoi(e) == 0); }();
  return on;
}
static bool epi_on() {        // GD_ROWS_GEMM_WS_EPI=0: the row-dots / rank-1 calls stay on the LDS-operand kernel (A/B switch)
  static const bool on = [] { const char* e = getenv("GD_ROWS_GEMM_WS_EPI"); return !(e && atoi(e) == 0); }();
  return on;
}
static int ws_min_rows() {
  static const int v = [] { const char* e = getenv("GD_ROWS_GEMM_WS_MIN_ROWS"); return e && atoi(e) > 0 ? atoi(e) : 65536; }();
  return v;
}

// -> GD_OK / error when the weight-stationary kernel took the call, 1 when it does not cover it (the caller goes on with
// the LDS-operand form).  Covered: widths in {64, 128}, bias only in the plain mode, no saved input, out not aliasing an input, enough rows that
// every wave gets units; ReLU on the input only in the plain / dots modes; the selector form dense or with an index list; row dots (u1 .. o2,
// no gate) behind a 128-wide input, the rank-1 + gate form (u1 .. o2 with gate_bits) in front of a 128-wide output.
int rows_gemm_ws_try(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel, const float* w, int32_t d_in, int32_t d_out,
                     int32_t trans_w, const float* bias, int32_t relu_in, const uint32_t* gate_bits, uint32_t* sign_out, float* out,
                     int64_t ld_out, float* save_in, void* stream, const float* in_alt, const uint8_t* sel, const float* u1,
                     const float* u2, float* o1, float* o2) {
  if (!ws_on() || matrix_split() != 0 || save_in || n_sel < ws_min_rows()) return 1;
  if (bias && (gate_bits || sign_out || u1 || !aligned16(bias) || !epi_on())) return 1;
  if (!((d_in == 64 || d_in == 128) && (d_out == 64 || d_out == 128))) return 1;
  if (in == out || in_alt == out || (relu_in && (gate_bits || sign_out)) || (sel && (gate_bits || sign_out))) return 1;
  if (sel && idx && u1 && d_in != 128) return 1;
  if (!aligned16(in) || !aligned16(out) || !aligned16(w) || (in_alt && !aligned16(in_alt)) || ld_in % 4 || ld_out % 4) return 1;
  if (u1 && (!u2 || !o1 || !o2 || sign_out || !aligned16(u1) || !aligned16(u2) || !epi_on())) return 1;
  hipStream_t s = (hipStream_t)stream;
  const WsEpi epi{u1, u2, o1, o2};
#define GD_WS_SHAPE(DI, DO) \
  return ws_dispatch<DI, DO>(in, ld_in, idx, n_sel, w, trans_w, relu_in, out, ld_out, gate_bits, sign_out, in_alt, sel, epi, bias, s)
  if (d_in == 128 && d_out == 128) GD_WS_SHAPE(128, 128);
  if (d_in == 128 && d_out == 64) GD_WS_SHAPE(128, 64);
  if (d_in == 64 && d_out == 128) GD_WS_SHAPE(64, 128);
  GD_WS_SHAPE(64, 64);
#undef GD_WS_SHAPE
}

}  // namespace gd

extern "C" int gd_rows_gemm_accumulate_f32(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel, const float* w,
                                           int32_t d_in, int32_t d_out, int32_t trans_w, float* out, int64_t ld_out, void* stream) {
  using namespace gd;
  GD_REQUIRE(in && w && out, GD_E_NULL, "gd_rows_gemm_accumulate_f32: null pointer");
  GD_REQUIRE(n_sel >= 0 && ld_in >= d_in && ld_out >= d_out, GD_E_DIM, "gd_rows_gemm_accumulate_f32: bad dims n_sel=%d d_in=%d d_out=%d", n_sel, d_in, d_out);
  if (n_sel == 0) return GD_OK;
  const int rc = rows_gemm_ws_accumulate(in, ld_in, idx, n_sel, w, d_in, d_out, trans_w, out, ld_out, stream);
  GD_REQUIRE(rc != 1, GD_E_DIM,
             "gd_rows_gemm_accumulate_f32: only where gd_rows_gemm_ws_covers(n_sel=%d, d_in=%d, d_out=%d) holds (16-byte aligned rows, in != out)",
             n_sel, d_in, d_out);
  return rc;
}

extern "C" int gd_rows_gemm_ws_covers(int32_t n_sel, int32_t d_in, int32_t d_out) {
  return gd::ws_on() && gd::matrix_split() == 0 && n_sel >= gd::ws_min_rows() && (d_in == 64 || d_in == 128) &&
                 (d_out == 64 || d_out == 128)
             ? 1
             : 0;
}
