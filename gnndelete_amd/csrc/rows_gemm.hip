// Row-subset GEMMs of the Del operator on the fp32 matrix cores.
//
//   forward / input-gradient :  out[r,:] = act(in[r,:]) @ W (or W^T),  r over an index list
//   weight gradient          :  dW = sum_s a[ia(s),:]^T g[ig(s),:]
//
// Both use v_mfma_f32_32x32x2_f32 (exact fp32 = a k-ordered fmaf chain, 64 FLOP/clk/SIMD: the
// Del GEMM at d=128 has intensity d/4 = 32 flop/B and sits on the compute side of the ridge).
//
// forward kernel: the [d_in, d_out] weight lives in LDS for the life of the block, interleaved by output tile
// (wl[k][r][t] = W(k, 32 t + r): the NT operands of a k step are one 16-byte read per lane); every wave owns
// one 32-row tile at a time and feeds its A operand STRAIGHT from global memory: lane l holds
// row (l&31) and loads 64 contiguous bytes per 32-wide k chunk = k-slots {32kc + 16(l>>5) + j},
// i.e. the k index of each MFMA step is permuted (both operands agree), which turns the gather
// into 16-byte loads that use every byte of the cache lines they touch, with no LDS round trip.  Output rows are written after the k loop, so `in` may alias `out`.
//
// wgrad kernel: the reduction runs over the selected rows; consecutive lanes read consecutive
// features of one row (128-byte coalesced), blocks own contiguous row chunks and write partial
// [d_a, d_b] products that a second kernel reduces in a fixed order (deterministic split-K).
#include "common.h"

namespace gd {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4v = __attribute__((ext_vector_type(4))) uint32_t;

// Split arithmetic (NP = 6 / 9 below): an fp32 value is the exact sum of three successive round-to-nearest bf16 pieces
// (3 x 8 significant bits + signs), x = x1 + x2 + x3, so the fp32 product s w is the sum of nine partial products
// s_i w_j, each EXACT in fp32 (8 x 8 bits), formed on v_mfma_f32_32x32x16_bf16 and accumulated in fp32 like the products
// of the fp32 instruction.  NP = 9 takes all nine; NP = 6 leaves out s2 w3, s3 w2, s3 w3 (each <= 2^-26 |s w|, a quarter
// of an fp32 rounding).  Measured against an fp64 product on the Del operator's shape (tools/experiments/split_lab.hip):
// rel-L2 1.6e-7 for both, 2.0e-7 for v_mfma_f32_32x32x2_f32 - and 44 us instead of 70 us: 2.7 x fewer matrix-pipe cycles
// (DESIGN section 4).
__device__ inline void split8(const float4 a, const float4 b, bf16x8 (&s)[3]) {
  const f32x2 v[4] = {{a.x, a.y}, {a.z, a.w}, {b.x, b.y}, {b.z, b.w}};
  u32x4v p[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x2 r = v[i];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));     // v_cvt_pk_bf16_f32 (RNE)
      p[q][i] = w;
      if (q < 2) {
        f32x2 h;
        h[0] = __builtin_bit_cast(float, w << 16);
        h[1] = __builtin_bit_cast(float, w & 0xffff0000u);
        r = r - h;                                           // exact
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) s[q] = __builtin_bit_cast(bf16x8, p[q]);
}

constexpr int kWgradMaxRows = 1024;   // rows of the reduction dimension one wgrad block owns

constexpr int kGemmThreads = 512;   // 8 waves share one LDS weight image; 2 blocks per CU

// optional: o1[row] = <out[row,:], u1>, o2[row] = <out[row,:], u2> from the epilogue (GAT's attention logits,
// which otherwise cost a second pass over the rows just written)
struct RowDots { const float* u1; const float* u2; float* o1; float* o2; };

// MODE 0: plain, 1: also emit the packed sign pattern of the output, 2: gate the output by such a pattern,
// 3: plain + the two row dot products of RowDots, 4: gated like 2 after a rank-1 correction of the product,
//    out[r, n] += o1[r] u1[n] + o2[r] u2[n] (RowDots reused: o1 / o2 per-row scalars READ by row id, u1 / u2 per column)
// SEL: rows may come from a second buffer (in_alt where sel[r] != 0); only instantiated for MODE 0 and 3
// NP: 0 = products on v_mfma_f32_32x32x2_f32; 6 / 9 = split arithmetic on the bf16 matrix instruction (above)
// KC (split form only): d_in / 32 as a compile-time constant - the wave keeps whole half rows in registers (wide outputs);
// KC = 0 with NP != 0: split products inside the chunk-wise loop of the fp32 form (narrow outputs: 4 waves per SIMD)
// QUEUE (fp32 form only): ONE block of 16 waves per CU shares the weight image and hands the tiles of the block's
// contiguous tile range out through an LDS counter.  The matrix pipe of a SIMD serves its resident waves oldest
// first (per-wave time stamps: four waves with identical static work finish at 109 / 136 / 156 / 166 us), so with a
// static tile-to-wave map the kernel lasts as long as its slowest wave; with the counter the waves that run ahead take
// more tiles and all end together (split_lab: 60.0 vs 61.8 us at the Del-1 size, 77.8 vs 80.4 us at N rows).
template <int NT, int MODE, bool SEL = false, int NP = 0, int KC = 0, bool QUEUE = false>
__global__ __launch_bounds__(QUEUE ? 1024 : kGemmThreads, (NP && KC) ? 2 : 4) void rows_gemm_mfma_kernel(
    const float* in, int64_t ld_in, const int32_t* __restrict__ idx, int32_t n_sel,
    const float* __restrict__ w, int32_t d_in, int32_t trans_w, const float* __restrict__ bias, int32_t relu_in,
    float* out, int64_t ld_out, float* __restrict__ save_in, const uint32_t* __restrict__ gate_bits,
    uint32_t* __restrict__ sign_out, const float* in_alt, const uint8_t* __restrict__ sel, RowDots dots) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  constexpr int d_out = 32 * NT;
  constexpr int kGemmThreads = QUEUE ? 1024 : gd::kGemmThreads;       // (shadows the namespace constant in here)
  constexpr int kWaves = kGemmThreads / 64;
  static_assert(!QUEUE || (NP == 0 && KC == 0), "the tile queue is built for the fp32 form");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // ---- weight image, interleaved by output tile: wl[(k * 32 + r) * NTP + t] = W(k, 32 t + r), so that the NT weight
  // operands one k step needs are ONE 16 / 8-byte LDS read per lane (conflict-free: consecutive lanes, consecutive
  // 16 bytes) instead of NT 4-byte reads - a third of the LDS instructions per MFMA of the row-major image.
  // Fill: one (k, r) pair per thread and step, its NT weights leave as one vector store.  With [k][n] weights
  // (trans_w = 0) the loads are coalesced as well; with [n][k] weights they are strided (slow fill: callers with
  // constant weights hand over a pre-transposed copy, see engine.py).
  constexpr int NTP = NT == 3 ? 4 : NT;
  // split form: three bf16 images [piece][k chunk kc][8-k group m][output tile t][khalf][feature r] of 16-byte operands
  // (k = 32 kc + 16 khalf + 8 m + c: the k slots lane (r, khalf) of the sample operand holds); one (8-k group, feature)
  // item per thread and step - 8 loads in flight, three 16-byte LDS stores
  bf16x8* const wimg = reinterpret_cast<bf16x8*>(wl);
  const int pstride = (d_in >> 3) * d_out;                   // operands per piece image
  if (NP) {
    for (int e = tid; e < pstride; e += kGemmThreads) {
      const int kg = e / d_out, n = e - kg * d_out;
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = trans_w ? w[(int64_t)n * d_in + 8 * kg + c] : w[(int64_t)(8 * kg + c) * d_out + n];
      bf16x8 pc[3];
      split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), pc);
      const int off = ((((kg >> 2) * 2 + (kg & 1)) * NT + (n >> 5)) * 2 + ((kg >> 1) & 1)) * 32 + (n & 31);
#pragma unroll
      for (int q = 0; q < 3; ++q) wimg[q * pstride + off] = pc[q];
    }
  }
  // (kFill steps of a thread requested together: a launch with a tile or two per wave - the Del products of a small request -
  //  lasts about as long as this fill, and a step-by-step fill is d_in * 32 / threads dependent round trips: 23 us for
  //  15,434 rows of a 128 x 128 product against 28 us for 60,000, tools/experiments/small_rows_time.py)
  constexpr int kFill = 8;                            // (one pass at d_in = 128)
  for (int e0 = tid; !NP && e0 < d_in * 32; e0 += kGemmThreads * kFill) {
    float v[kFill][NTP];
#pragma unroll
    for (int u = 0; u < kFill; ++u) {
      const int e = min(e0 + u * kGemmThreads, d_in * 32 - 1);
      const int k = e >> 5, r = e & 31;
#pragma unroll
      for (int t = 0; t < NTP; ++t)
        v[u][t] = t < NT ? (trans_w ? w[(int64_t)(32 * t + r) * d_in + k] : w[(int64_t)k * d_out + 32 * t + r]) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kFill; ++u) {
      const int e = e0 + u * kGemmThreads;
      if (e < d_in * 32) {
        if (NTP == 4) *reinterpret_cast<float4*>(wl + e * 4) = make_float4(v[u][0], v[u][1], v[u][2], v[u][3 % NTP]);
        else if (NTP == 2) *reinterpret_cast<float2*>(wl + e * 2) = make_float2(v[u][0], v[u][1 % NTP]);
        else wl[e] = v[u][0];
      }
    }
  }
  __syncthreads();

  const int n_tiles = (n_sel + 31) >> 5;
  const int r_lo = lane & 31;       // the sample (row of `in`) this lane owns
  const int khalf = lane >> 5;      // which 16-float half of each 32-wide k chunk
  const int stride = gridDim.x * kWaves;
  const int kchunks = d_in >> 5;

  // Software pipeline over chunks AND tiles: while chunk kc feeds the matrix cores, chunk kc+1 -
  // or chunk 0 of this wave's NEXT tile - is already in flight, so a tile never starts with an
  // exposed index + HBM round trip.  All loads are unconditional (clamped rows), no branches.
  auto row_of = [&](int tile_) -> int32_t {       // 32-bit row ids: registers are the scarce resource here
    const int s_ = min(tile_ * 32 + r_lo, n_sel - 1);
    return idx ? idx[s_] : s_;
  };
  // row r is read from in_alt instead of in where sel[r] != 0 (a matrix whose rows live in two buffers)
  auto base_of = [&](int32_t r) -> const float* { return (SEL && sel[r]) ? in_alt : in; };
  // ---- epilogue: D[i][j], j = lane&31 = sample, i = (r&3) + 8*(r>>2) + 4*(lane>>5) = feature
  auto epilogue = [&](f32x16 (&acc)[NT], int32_t row_cur, int s_a, bool live, const uint32_t (&gate_w)[NT], float r1a, float r1b) {
      // (feature 32t + 8q + 4 khalf + c sits in acc[t][4q + c]; bit b of word t of a row's packed
      //  sign / gate mask is feature 32t + b, so this lane owns bits 8q + 4 khalf + c of each word)
      float* dst = out + (int64_t)row_cur * ld_out + 4 * khalf;
      constexpr bool want_dots = MODE == 3;
      // (the u vectors are the same for every tile: without the clobber LICM keeps all of them in registers)
      if (want_dots) asm volatile("" ::: "memory");
      float dot1 = 0.f, dot2 = 0.f;
  #pragma unroll
      for (int t = 0; t < NT; ++t) {
        uint32_t pos = 0;
  #pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 v = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
          const int n0 = 32 * t + 8 * q;
          if (bias) v = f4_add(v, *reinterpret_cast<const float4*>(bias + n0 + 4 * khalf));
          if (want_dots) {
            asm volatile("" ::: "memory");      // one pair of u loads in flight, not all 8 NT of them
            const float4 p1 = *reinterpret_cast<const float4*>(dots.u1 + n0 + 4 * khalf);
            const float4 p2 = *reinterpret_cast<const float4*>(dots.u2 + n0 + 4 * khalf);
            dot1 = fmaf(v.x, p1.x, dot1); dot1 = fmaf(v.y, p1.y, dot1); dot1 = fmaf(v.z, p1.z, dot1); dot1 = fmaf(v.w, p1.w, dot1);
            dot2 = fmaf(v.x, p2.x, dot2); dot2 = fmaf(v.y, p2.y, dot2); dot2 = fmaf(v.z, p2.z, dot2); dot2 = fmaf(v.w, p2.w, dot2);
          }
          if (MODE == 4) {
            const float4 p1 = *reinterpret_cast<const float4*>(dots.u1 + n0 + 4 * khalf);
            const float4 p2 = *reinterpret_cast<const float4*>(dots.u2 + n0 + 4 * khalf);
            v.x = fmaf(r1b, p2.x, fmaf(r1a, p1.x, v.x)); v.y = fmaf(r1b, p2.y, fmaf(r1a, p1.y, v.y));
            v.z = fmaf(r1b, p2.z, fmaf(r1a, p1.z, v.z)); v.w = fmaf(r1b, p2.w, fmaf(r1a, p1.w, v.w));
          }
          if (MODE == 2 || MODE == 4) {   // ReLU backward: pass the gradient where the forward activation input was > 0
            const uint32_t m = gate_w[t] >> (8 * q + 4 * khalf);
            v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f;
            v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
          }
          if (MODE == 1)
            pos |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u))
                   << (8 * q + 4 * khalf);
          if (live) *reinterpret_cast<float4*>(dst + n0) = v;
        }
        if (MODE == 1) {     // the two half-row lanes merge their bits (all lanes shuffle)
          pos |= (uint32_t)__shfl_xor((int)pos, 32);
          if (live && khalf == 0) sign_out[(int64_t)s_a * NT + t] = pos;
        }
      }
      if (want_dots) {       // the two half-row lanes of a sample add their halves (all lanes shuffle)
        dot1 += __shfl_xor(dot1, 32);
        dot2 += __shfl_xor(dot2, 32);
        if (live && khalf == 0) { dots.o1[row_cur] = dot1; dots.o2[row_cur] = dot2; }
      }
  };

  if constexpr (NP != 0 && KC != 0) {
    // ---- split form: whole half rows in registers (4 KC float4 per lane), the NEXT tile's in flight while this one feeds
    // the matrix cores - a 32-wide k chunk lasts < 1 us here, chunk-wise prefetch no longer covers the HBM latency
    int tile = wave * gridDim.x + blockIdx.x;          // (wave-major: see the static hand-out below)
    if (tile >= n_tiles) return;
    auto fetch = [&](int t_, float4 (&a)[4 * KC], int32_t& row) {
      row = row_of(min(t_, n_tiles - 1));
      const float4* s0 = reinterpret_cast<const float4*>(base_of(row) + (int64_t)row * ld_in) + khalf * 4;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[kc * 4 + i] = s0[kc * 8 + i];
    };
    auto work = [&](int t_, float4 (&a)[4 * KC], int32_t row) {
      const int s_a = t_ * 32 + r_lo;
      const bool live = s_a < n_sel;
      uint32_t gate_w[NT];
      float r1a = 0.f, r1b = 0.f;
      if (MODE == 2 || MODE == 4) {
        const uint32_t* gsrc = gate_bits + (int64_t)min(s_a, n_sel - 1) * NT;
#pragma unroll
        for (int t = 0; t < NT; ++t) gate_w[t] = gsrc[t];
        if (MODE == 4) { r1a = dots.o1[row]; r1b = dots.o2[row]; }
      }
      if (save_in && live) {
        float4* sav = reinterpret_cast<float4*>(save_in + (int64_t)s_a * d_in) + khalf * 4;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
          for (int i = 0; i < 4; ++i) sav[kc * 8 + i] = a[kc * 4 + i];
      }
      f32x16 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          float4 lo = a[kc * 4 + 2 * m], hi = a[kc * 4 + 2 * m + 1];
          if (relu_in) {
            lo.x = fmaxf(lo.x, 0.f); lo.y = fmaxf(lo.y, 0.f); lo.z = fmaxf(lo.z, 0.f); lo.w = fmaxf(lo.w, 0.f);
            hi.x = fmaxf(hi.x, 0.f); hi.y = fmaxf(hi.y, 0.f); hi.z = fmaxf(hi.z, 0.f); hi.w = fmaxf(hi.w, 0.f);
          }
          bf16x8 sp[3];
          split8(lo, hi, sp);
          // transposed product (weight = "A" operand: D rows = output features; sample = "B" operand: D cols = samples,
          // as in the fp32 form below), small terms first.  Two output tiles at a time: consecutive matrix instructions
          // accumulate into different registers, so the wave issues them back to back instead of waiting out each
          // one's latency on a six-deep dependent chain
#pragma unroll
          for (int t0 = 0; t0 < NT; t0 += 2) {
            constexpr int kPairs[9][2] = {{2, 2}, {2, 1}, {1, 2}, {2, 0}, {1, 1}, {0, 2}, {1, 0}, {0, 1}, {0, 0}};   // (weight piece, sample piece)
            bf16x8 wq[2][3];
#pragma unroll
            for (int d = 0; d < 2; ++d)
              if (t0 + d < NT) {
                const bf16x8* wk = wimg + (((kc * 2 + m) * NT + t0 + d) * 2 + khalf) * 32 + r_lo;
                wq[d][0] = wk[0]; wq[d][1] = wk[pstride]; wq[d][2] = wk[2 * pstride];
              }
#pragma unroll
            for (int pr = (NP == 9 ? 0 : 3); pr < 9; ++pr)
#pragma unroll
              for (int d = 0; d < 2; ++d)
                if (t0 + d < NT)
                  acc[t0 + d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[d][kPairs[pr][0]], sp[kPairs[pr][1]], acc[t0 + d], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);                 // keep the LDS reads of later k groups where they are
        }
      }
      epilogue(acc, row, s_a, live, gate_w, r1a, r1b);
    };
    constexpr int KCR = KC ? KC : 1;
    float4 ra[4 * KCR], rb[4 * KCR];
    int32_t rowa, rowb;
    fetch(tile, ra, rowa);
    for (; tile < n_tiles; tile += 2 * stride) {
      fetch(tile + stride, rb, rowb);
      work(tile, ra, rowa);
      if (tile + stride >= n_tiles) break;
      fetch(tile + 2 * stride, ra, rowa);
      work(tile + stride, rb, rowb);
    }
    return;
  }

  // tile hand-out: static (tile = wave * blocks + block, + stride) or, with QUEUE, tickets from the block's LDS counter
  // over its contiguous range [t_lo, t_hi) - a ticket past the range ends the wave (n_tiles = "none")
  __shared__ int q_next;
  const int per_block = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_lo = blockIdx.x * per_block, t_hi = min(n_tiles, t_lo + per_block);
  if (QUEUE) {
    if (tid == 0) q_next = 0;
    __syncthreads();
  }
  auto grab = [&]() -> int {
    int t = 0;
    if (lane == 0) t = atomicAdd(&q_next, 1);
    t = __builtin_amdgcn_readfirstlane(t) + t_lo;
    return t < t_hi ? t : n_tiles;
  };
  // (static: WAVE-major - tile = wave * blocks + block - so that a launch with fewer tiles than wave slots spreads them over the
  //  compute units, one wave per SIMD first: a 128 x 128 tile is 6.8 us of matrix instructions, and 483 tiles on 61 blocks of
  //  eight waves ran two per SIMD on a quarter of the chip - 23 us for 15,434 rows against 28 us for 60,000)
  int tile = QUEUE ? grab() : wave * (int)gridDim.x + (int)blockIdx.x;
  if (tile >= n_tiles) return;
  int tile_nxt = QUEUE ? grab() : tile + stride;
  int32_t row_cur = row_of(tile);
  int32_t row_nxt = row_of(min(tile_nxt, n_tiles - 1));
  float4 a_next[4];
  {
    const float4* src0 = reinterpret_cast<const float4*>(base_of(row_cur) + (int64_t)row_cur * ld_in) + khalf * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) a_next[i] = src0[i];
  }
  for (; tile < n_tiles;) {
    const int s_a = tile * 32 + r_lo;
    const bool live = s_a < n_sel;
    const float4* src = reinterpret_cast<const float4*>(base_of(row_cur) + (int64_t)row_cur * ld_in) + khalf * 4;
    const float4* src_n = reinterpret_cast<const float4*>(base_of(row_nxt) + (int64_t)row_nxt * ld_in) + khalf * 4;
    float4* sav = save_in ? reinterpret_cast<float4*>(save_in + (int64_t)s_a * d_in) + khalf * 4 : nullptr;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // packed ReLU gate of this lane's row (one bit per output feature), in flight during the k loop
    uint32_t gate_w[NT];
    float r1a = 0.f, r1b = 0.f;                  // MODE 4: this row's two rank-1 coefficients
    if (MODE == 2 || MODE == 4) {
      const uint32_t* gsrc = gate_bits + (int64_t)min(s_a, n_sel - 1) * NT;
#pragma unroll
      for (int t = 0; t < NT; ++t) gate_w[t] = gsrc[t];
      if (MODE == 4) { r1a = dots.o1[row_cur]; r1b = dots.o2[row_cur]; }
    }

    for (int kc = 0; kc < kchunks; ++kc) {
      float4 a4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a4[i] = a_next[i];
      const float4* nsrc = (kc + 1 < kchunks) ? src + (kc + 1) * 8 : src_n;
#pragma unroll
      for (int i = 0; i < 4; ++i) a_next[i] = nsrc[i];
      if (sav && live) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sav[kc * 8 + i] = a4[i];
      }
      if (NP) {                                             // split products, chunk by chunk (KC == 0)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (relu_in) {
            a4[i].x = fmaxf(a4[i].x, 0.f); a4[i].y = fmaxf(a4[i].y, 0.f);
            a4[i].z = fmaxf(a4[i].z, 0.f); a4[i].w = fmaxf(a4[i].w, 0.f);
          }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          bf16x8 sp[3];
          split8(a4[2 * m], a4[2 * m + 1], sp);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const bf16x8* wk = wimg + (((kc * 2 + m) * NT + t) * 2 + khalf) * 32 + r_lo;
            const bf16x8 w1 = wk[0], w2 = wk[pstride], w3 = wk[2 * pstride];
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, sp[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, sp[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, sp[2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, sp[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, sp[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, sp[0], acc[t], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int i = 0; !NP && i < 4; ++i) {
        if (relu_in) {
          a4[i].x = fmaxf(a4[i].x, 0.f); a4[i].y = fmaxf(a4[i].y, 0.f);
          a4[i].z = fmaxf(a4[i].z, 0.f); a4[i].w = fmaxf(a4[i].w, 0.f);
        }
        const float av[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
        const int k0 = kc * 32 + khalf * 16 + i * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float* wk = wl + ((k0 + s) * 32 + r_lo) * NTP;
          float wv[NTP];
          if (NTP == 4) {
            const float4 f = *reinterpret_cast<const float4*>(wk);
            wv[0] = f.x; wv[1 % NTP] = f.y; wv[2 % NTP] = f.z; wv[3 % NTP] = f.w;
          } else if (NTP == 2) {
            const float2 f = *reinterpret_cast<const float2*>(wk);
            wv[0] = f.x; wv[1 % NTP] = f.y;
          } else {
            wv[0] = wk[0];
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            // transposed product: the weight is the MFMA "A" operand (D rows = output features),
            // the sample row the "B" operand (D cols = samples), so every lane ends up holding
            // 4-float runs of ITS OWN output row -> float4 stores, no index reload
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t], av[s], acc[t], 0, 0, 0);
          }
        }
      }
    }

    epilogue(acc, row_cur, s_a, live, gate_w, r1a, r1b);
    row_cur = row_nxt;
    tile = tile_nxt;
    tile_nxt = QUEUE ? (tile < n_tiles ? grab() : n_tiles) : tile + stride;
    // row index of the tile after next: its first loads are only issued at the end of the next tile
    row_nxt = row_of(min(tile_nxt, n_tiles - 1));
  }
}

// Fallback for dimensions the MFMA path does not cover: one wave per selected row, the row is
// held in registers (so in/out may alias), one output column at a time.
__global__ __launch_bounds__(256) void rows_gemm_scalar_kernel(
    const float* in, int64_t ld_in, const int32_t* __restrict__ idx, int32_t n_sel,
    const float* __restrict__ w, int32_t d_in, int32_t d_out, int32_t trans_w, const float* __restrict__ bias,
    int32_t relu_in, float* out, int64_t ld_out, float* __restrict__ save_in, const uint32_t* __restrict__ gate_bits,
    uint32_t* __restrict__ sign_out, const float* in_alt, const uint8_t* __restrict__ sel) {
  constexpr int kMaxPerLane = 16;  // d_in <= 1024
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n_sel) return;
  const int64_t row = idx ? idx[s] : s;
  float xr[kMaxPerLane];
#pragma unroll
  for (int q = 0; q < kMaxPerLane; ++q) {
    const int k = lane + q * kWave;
    float v = k < d_in ? ((sel && sel[row]) ? in_alt : in)[row * ld_in + k] : 0.f;
    if (save_in && k < d_in) save_in[(int64_t)s * d_in + k] = v;
    xr[q] = relu_in ? fmaxf(v, 0.f) : v;
  }
  const int n_words = (d_out + 31) >> 5;
  uint32_t word = 0;
  for (int j = 0; j < d_out; ++j) {
    float p = 0.f;
#pragma unroll
    for (int q = 0; q < kMaxPerLane; ++q) {
      const int k = lane + q * kWave;
      if (k < d_in) p = fmaf(xr[q], trans_w ? w[(int64_t)j * d_in + k] : w[(int64_t)k * d_out + j], p);
    }
    p = wave_sum(p);
    p += bias ? bias[j] : 0.f;
    if (gate_bits && !((gate_bits[(int64_t)s * n_words + (j >> 5)] >> (j & 31)) & 1u)) p = 0.f;
    if (p > 0.f) word |= 1u << (j & 31);
    if (lane == 0) {
      out[row * ld_out + j] = p;
      if (sign_out && ((j & 31) == 31 || j == d_out - 1)) sign_out[(int64_t)s * n_words + (j >> 5)] = word;
    }
    if ((j & 31) == 31) word = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient:  dW[32 TA, 32 TB] = sum_s a[ia(s),:]^T g[ig(s),:]
// The reduction dimension is the (gathered) row index.  A block streams tiles of 32 rows of both
// operands through LDS (float4 global loads, double-buffered: the loads of tile t+1 are in flight
// while tile t feeds the matrix cores); its 4 waves split the TA x TB output tiles and read their
// MFMA operands straight out of the row-major tiles (lane -> consecutive column: conflict-free).
// HBM bytes S (d_a + d_b) 4 and flops 2 S d_a d_b are balanced at d = 128 (ridge ~20 flop/B).
// NW = waves per block (they split the TA x TB output tiles).
//
// LOSS: the upstream gradient is not read but FORMED while it is fetched, from the folded row-target MSE
// terms of that layer (loss.hip): g = coef_u (z - tbar_u) for a row with loss slot u, 0 otherwise
// (+ g_add).  The loss kernel, its [S, d] gradient write and this kernel's read of it disappear; the
// two loss sums (cnt_u |z - tbar_u|^2 per kind) fall out of the same fetch as per-block partials.
struct WgradLoss {
  const int32_t* slot;     // [n_sel]: loss slot of selected row s, or -1
  const float* tm;         // [n_slots, d_b] mean targets (compact)
  const float* coef;       // [n_slots] gradient scale 2 w c
  const float* cnt_signed; // [n_slots] term count, NI terms stored negative (kind folded into the sign)
  float* partials;         // [2 * n_blocks] per-block (DEC, NI) sums of cnt |z - tbar|^2
};

template <int TA, int TB, bool LOSS, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 4) void rows_wgrad_mfma_kernel(
    const float* __restrict__ a, int64_t ld_a, const int32_t* __restrict__ a_idx, const float* __restrict__ g,
    int64_t ld_g, const int32_t* __restrict__ g_idx, const float* __restrict__ relu_mask,
    const float* __restrict__ g_add, int32_t n_sel, int32_t rows_per_block, float* __restrict__ partials,
    WgradLoss loss) {
  constexpr int DA = 32 * TA, DB = 32 * TB, KT = 32;
  constexpr int TILES = TA * TB, TPW = (TILES + NW - 1) / NW;
  constexpr int NTH = 64 * NW;
  constexpr int FA = DA / 4, FB = DB / 4;            // float4 per row
  constexpr int LA = (KT * FA + NTH - 1) / NTH, LB = (KT * FB + NTH - 1) / NTH;  // float4 loads per thread per tile
  __shared__ __attribute__((aligned(16))) float sa[2][KT * DA];
  __shared__ __attribute__((aligned(16))) float sg[2][KT * DB];
  __shared__ int32_t sia[kWgradMaxRows], sig[kWgradMaxRows];   // this block's gather lists
  __shared__ int32_t sls[LOSS ? kWgradMaxRows : 1];            // and loss slots
  __shared__ float lred[2][NW];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int c_lo = lane & 31, khalf = lane >> 5;
  float ls0 = 0.f, ls1 = 0.f;

  f32x16 acc[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int s_begin = blockIdx.x * rows_per_block;
  const int s_end = min(n_sel, s_begin + rows_per_block);
  const int n_tiles = (s_end - s_begin + KT - 1) / KT;

  // stage the row indices once so the per-tile row loads do not wait on an index load
  for (int i = tid; i < s_end - s_begin; i += NTH) {
    sia[i] = a_idx ? a_idx[s_begin + i] : s_begin + i;
    sig[i] = g_idx ? g_idx[s_begin + i] : s_begin + i;
    if (LOSS) sls[i] = loss.slot[s_begin + i];
  }
  __syncthreads();

  // The operand loads of tile t+1 are ISSUED before tile t feeds the matrix cores and only COMBINED (loss gradient,
  // ReLU mask, g_add) after it, in finish(): nothing between the two may touch the loaded registers, or the
  // compiler has to drain the loads before the MFMA loop and their whole latency is exposed once per tile
  // (measured: 74 us with two raw operands, 106 us when the loss terms were formed inside the fetch).
  float4 ra[LA], rg[LB], rt[LB], rx[LB];      // a rows | g (or z) rows | folded targets (LOSS) or ReLU mask | g_add
  float rcf[LB], rcn[LB];
  auto fetch = [&](int tile) {
    const int s0 = s_begin + tile * KT;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int f = tid + NTH * i, r = f / FA, c4 = f % FA, ss = s0 + r;
      ra[i] = f4_zero();
      if (ss < s_end && f < KT * FA) ra[i] = reinterpret_cast<const float4*>(a + (int64_t)sia[ss - s_begin] * ld_a)[c4];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int f = tid + NTH * i, r = f / FB, c4 = f % FB, ss = s0 + r;
      rg[i] = f4_zero();
      rx[i] = f4_zero();
      rt[i] = LOSS ? f4_zero() : make_float4(1.f, 1.f, 1.f, 1.f);      // (a mask of ones = no mask)
      rcf[i] = 0.f;
      rcn[i] = 0.f;
      if (ss < s_end && f < KT * FB) {
        const int64_t row = sig[ss - s_begin];
        if (LOSS) {
          // branch-free: a row without loss terms (slot -1) reads slot 0's target with coefficient and count 0
          const int u = sls[ss - s_begin], uu = max(u, 0);
          rg[i] = reinterpret_cast<const float4*>(g + row * ld_g)[c4];
          rt[i] = reinterpret_cast<const float4*>(loss.tm + (int64_t)uu * DB)[c4];
          rcf[i] = u >= 0 ? loss.coef[uu] : 0.f;
          rcn[i] = u >= 0 ? loss.cnt_signed[uu] : 0.f;
        } else {
          rg[i] = reinterpret_cast<const float4*>(g + row * ld_g)[c4];
          if (relu_mask) rt[i] = reinterpret_cast<const float4*>(relu_mask + row * ld_g)[c4];
        }
        if (g_add) rx[i] = reinterpret_cast<const float4*>(g_add + row * ld_g)[c4];
      }
    }
  };
  // rg <- the upstream gradient rows of the fetched tile (same arithmetic, same order as before the split)
  auto finish = [&]() {
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      float4 v;
      if (LOSS) {
        const float4 zv = rg[i], tv = rt[i];
        const float cf = rcf[i], cn = rcn[i];
        const float4 df = make_float4(zv.x - tv.x, zv.y - tv.y, zv.z - tv.z, zv.w - tv.w);
        float sq = df.x * df.x;
        sq = fmaf(df.y, df.y, sq); sq = fmaf(df.z, df.z, sq); sq = fmaf(df.w, df.w, sq);
        if (cn >= 0.f) ls0 = fmaf(cn, sq, ls0); else ls1 = fmaf(-cn, sq, ls1);
        v = make_float4(cf * df.x, cf * df.y, cf * df.z, cf * df.w);
      } else {
        v = rg[i];
        if (relu_mask) {
          const float4 m = rt[i];
          v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
          v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
      }
      if (g_add) v = f4_add(v, rx[i]);
      rg[i] = v;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LA; ++i)
      if (tid + NTH * i < KT * FA) reinterpret_cast<float4*>(sa[buf])[tid + NTH * i] = ra[i];
#pragma unroll
    for (int i = 0; i < LB; ++i)
      if (tid + NTH * i < KT * FB) reinterpret_cast<float4*>(sg[buf])[tid + NTH * i] = rg[i];
  };

  if (n_tiles > 0) {
    fetch(0);
    finish();
    stash(0);
  }
  __syncthreads();
  for (int tile = 0; tile < n_tiles; ++tile) {
    const int cur = tile & 1;
    if (tile + 1 < n_tiles) fetch(tile + 1);
    const float* pa = sa[cur];
    const float* pg = sg[cur];
#pragma unroll 4
    for (int kk = 0; kk < KT / 2; ++kk) {
      const int k = 2 * kk + khalf;
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        const int id = wave + NW * q;
        if (TILES % NW == 0 || id < TILES) {
          const float av = pa[k * DA + (id / TB) * 32 + c_lo];
          const float gv = pg[k * DB + (id % TB) * 32 + c_lo];
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, gv, acc[q], 0, 0, 0);
        }
      }
    }
    if (tile + 1 < n_tiles) {
      finish();
      stash(cur ^ 1);
    }
    __syncthreads();
  }

  float* dst = partials + (int64_t)blockIdx.x * DA * DB;
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int id = wave + NW * q;
    if (id >= TILES) continue;
    const int i0 = (id / TB) * 32, j0 = (id % TB) * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2) + 4 * khalf;
      dst[(int64_t)(i0 + rr) * DB + j0 + c_lo] = acc[q][r];
    }
  }
  if (LOSS) {
    ls0 = wave_sum(ls0);
    ls1 = wave_sum(ls1);
    if (lane == 0) { lred[0][wave] = ls0; lred[1][wave] = ls1; }
    __syncthreads();
    if (tid == 0) {
      float p0 = 0.f, p1 = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) { p0 += lred[0][q]; p1 += lred[1][q]; }
      loss.partials[2 * blockIdx.x + 0] = p0;
      loss.partials[2 * blockIdx.x + 1] = p1;
    }
  }
}

// generic dims: one thread per dW element, loop over the block's rows
__global__ __launch_bounds__(256) void rows_wgrad_scalar_kernel(
    const float* __restrict__ a, int64_t ld_a, const int32_t* __restrict__ a_idx, const float* __restrict__ g,
    int64_t ld_g, const int32_t* __restrict__ g_idx, const float* __restrict__ relu_mask,
    const float* __restrict__ g_add, int32_t n_sel, int32_t d_a, int32_t d_b, int32_t rows_per_block,
    float* __restrict__ partials) {
  const int s_begin = blockIdx.x * rows_per_block;
  const int s_end = min(n_sel, s_begin + rows_per_block);
  float* dst = partials + (int64_t)blockIdx.x * d_a * d_b;
  for (int e = threadIdx.x; e < d_a * d_b; e += 256) {
    const int i = e / d_b, j = e % d_b;
    float p = 0.f;
    for (int s = s_begin; s < s_end; ++s) {
      const int64_t ra = a_idx ? a_idx[s] : s, rg = g_idx ? g_idx[s] : s;
      float gg = g[rg * ld_g + j];
      if (relu_mask) gg = relu_mask[rg * ld_g + j] > 0.f ? gg : 0.f;
      if (g_add) gg += g_add[rg * ld_g + j];
      p = fmaf(a[ra * ld_a + i], gg, p);
    }
    dst[e] = p;
  }
}

// dW[e] = (accumulate ? dW[e] : 0) + sum_b partials[b][e], summed in a fixed order: a block owns
// 64 consecutive elements (16 float4 lanes) x 16 interleaved slices of the partial list, then
// folds the 16 slice sums through LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partials, int32_t n_part,
                                                           int32_t n_elem, int32_t accumulate,
                                                           float* __restrict__ dw) {
  __shared__ float4 red[16][16];
  const int v = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int e4 = blockIdx.x * 16 + v;                     // float4 index
  const int n4 = n_elem >> 2;
  // same association as wgrad_reduce_adam_kernel (the two must agree bit for bit)
  float4 s0 = f4_zero(), s1 = f4_zero(), s2 = f4_zero(), s3 = f4_zero();
  if (e4 < n4) {
    int b = slice;
    for (; b + 48 < n_part; b += 64) {
      const float4 p0 = reinterpret_cast<const float4*>(partials + (int64_t)b * n_elem)[e4];
      const float4 p1 = reinterpret_cast<const float4*>(partials + (int64_t)(b + 16) * n_elem)[e4];
      const float4 p2 = reinterpret_cast<const float4*>(partials + (int64_t)(b + 32) * n_elem)[e4];
      const float4 p3 = reinterpret_cast<const float4*>(partials + (int64_t)(b + 48) * n_elem)[e4];
      s0 = f4_add(s0, p0); s1 = f4_add(s1, p1); s2 = f4_add(s2, p2); s3 = f4_add(s3, p3);
    }
    for (; b < n_part; b += 16) s0 = f4_add(s0, reinterpret_cast<const float4*>(partials + (int64_t)b * n_elem)[e4]);
  }
  red[slice][v] = f4_add(f4_add(s0, s1), f4_add(s2, s3));
  __syncthreads();
  if (slice == 0 && e4 < n4) {
    float4 t = accumulate ? reinterpret_cast<float4*>(dw)[e4] : f4_zero();
#pragma unroll
    for (int i = 0; i < 16; ++i) t = f4_add(t, red[i][v]);
    reinterpret_cast<float4*>(dw)[e4] = t;
  }
}

// Same reduction with torch.optim.Adam applied to the freshly reduced gradient in the epilogue
// (t = *iter + 1 from the shared iteration counter): saves the separate optimizer launch.
__global__ __launch_bounds__(256) void wgrad_reduce_adam_kernel(const float* __restrict__ partials, int32_t n_part,
                                                                int32_t n_elem, int32_t accumulate,
                                                                float* __restrict__ dw, float* __restrict__ param,
                                                                float* __restrict__ m, float* __restrict__ v,
                                                                const int32_t* __restrict__ iter, double lr, double beta1,
                                                                double beta2, double eps) {
  __shared__ float4 red[16][16];
  const int vv = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int e4 = blockIdx.x * 16 + vv;
  const int n4 = n_elem >> 2;
  // four partial rows in flight per thread (fixed association: s0 + s1 + s2 + s3 of the slice, then the slices)
  float4 s0 = f4_zero(), s1 = f4_zero(), s2 = f4_zero(), s3 = f4_zero();
  if (e4 < n4) {
    int b = slice;
    for (; b + 48 < n_part; b += 64) {
      const float4 p0 = reinterpret_cast<const float4*>(partials + (int64_t)b * n_elem)[e4];
      const float4 p1 = reinterpret_cast<const float4*>(partials + (int64_t)(b + 16) * n_elem)[e4];
      const float4 p2 = reinterpret_cast<const float4*>(partials + (int64_t)(b + 32) * n_elem)[e4];
      const float4 p3 = reinterpret_cast<const float4*>(partials + (int64_t)(b + 48) * n_elem)[e4];
      s0 = f4_add(s0, p0); s1 = f4_add(s1, p1); s2 = f4_add(s2, p2); s3 = f4_add(s3, p3);
    }
    for (; b < n_part; b += 16) s0 = f4_add(s0, reinterpret_cast<const float4*>(partials + (int64_t)b * n_elem)[e4]);
  }
  red[slice][vv] = f4_add(f4_add(s0, s1), f4_add(s2, s3));
  __syncthreads();
  if (slice == 0 && e4 < n4) {
    float4 t4 = accumulate ? reinterpret_cast<float4*>(dw)[e4] : f4_zero();
#pragma unroll
    for (int i = 0; i < 16; ++i) t4 = f4_add(t4, red[i][vv]);
    reinterpret_cast<float4*>(dw)[e4] = t4;
    const AdamScalars sc = adam_scalars(lr, beta1, beta2, eps, *iter + 1);
    float4 p4 = reinterpret_cast<float4*>(param)[e4], m4 = reinterpret_cast<float4*>(m)[e4],
           v4 = reinterpret_cast<float4*>(v)[e4];
    adam_update(p4.x, m4.x, v4.x, t4.x, sc);
    adam_update(p4.y, m4.y, v4.y, t4.y, sc);
    adam_update(p4.z, m4.z, v4.z, t4.z, sc);
    adam_update(p4.w, m4.w, v4.w, t4.w, sc);
    reinterpret_cast<float4*>(param)[e4] = p4;
    reinterpret_cast<float4*>(m)[e4] = m4;
    reinterpret_cast<float4*>(v)[e4] = v4;
  }
}

// The launch-sized tail of a training step in ONE launch: the split-K reductions of both Del weights (block order, the
// association of wgrad_reduce_adam_kernel - bit-identical results) each with its Adam update, and the loss finalize (per-block
// loss partials -> history ring, ring position, iteration counter).  Three launches and two kernel boundaries less per step.
// The Adam blocks need the step number t = *iter + 1 and the finalize block advances *iter in the same launch: every Adam block
// reads the counter FIRST (a returning device-scope atomic) and then checks in on `arrive`; the finalize thread writes the counter
// only after all of them have checked in, and resets `arrive` for the next launch.  The grid (n_elem / 64 blocks per weight + 1:
// 321 at 128 x 128 + 64 x 64) is a fraction of the resident set, so every block is scheduled while the one thread waits.
struct TailJob {
  const float* partials; int32_t n_part, n_elem, accumulate;
  float* dw; float* param; float* m; float* v;
};
struct TailFin {
  const float* p1; int32_t n1; const float* p2; int32_t n2;
  float* hist; int32_t capacity; int32_t* pos; int32_t* iter; int32_t* arrive;
};

__global__ __launch_bounds__(256) void step_tail_kernel(TailJob j1, TailJob j2, TailFin fin, double lr, double beta1, double beta2,
                                                        double eps) {
  __shared__ float4 red[16][16];
  __shared__ float fred[4][256];
  __shared__ int32_t t_sh;
  const int nb1 = (j1.n_elem / 4 + 15) / 16, nb2 = (j2.n_elem / 4 + 15) / 16;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < nb1 + nb2) {
    const bool first = (int)blockIdx.x < nb1;
    const TailJob j = first ? j1 : j2;
    const int blk = first ? blockIdx.x : blockIdx.x - nb1;
    if (tid == 0) {
      int32_t t = atomicAdd(fin.iter, 0);                 // the counter as the L2 holds it, before this block checks in
      asm volatile("" : "+v"(t));
      atomicAdd(fin.arrive + (t < 0 ? 1 : 0), 1);         // (address depends on t: the check-in cannot pass the read)
      t_sh = t + 1;
    }
    const int vv = tid & 15, slice = tid >> 4;
    const int e4 = blk * 16 + vv;
    const int n4 = j.n_elem >> 2;
    float4 s0 = f4_zero(), s1 = f4_zero(), s2 = f4_zero(), s3 = f4_zero();
    if (e4 < n4) {
      int b = slice;
      for (; b + 48 < j.n_part; b += 64) {
        const float4 p0 = reinterpret_cast<const float4*>(j.partials + (int64_t)b * j.n_elem)[e4];
        const float4 p1 = reinterpret_cast<const float4*>(j.partials + (int64_t)(b + 16) * j.n_elem)[e4];
        const float4 p2 = reinterpret_cast<const float4*>(j.partials + (int64_t)(b + 32) * j.n_elem)[e4];
        const float4 p3 = reinterpret_cast<const float4*>(j.partials + (int64_t)(b + 48) * j.n_elem)[e4];
        s0 = f4_add(s0, p0); s1 = f4_add(s1, p1); s2 = f4_add(s2, p2); s3 = f4_add(s3, p3);
      }
      for (; b < j.n_part; b += 16) s0 = f4_add(s0, reinterpret_cast<const float4*>(j.partials + (int64_t)b * j.n_elem)[e4]);
    }
    red[slice][vv] = f4_add(f4_add(s0, s1), f4_add(s2, s3));
    __syncthreads();
    if (slice == 0 && e4 < n4) {
      float4 t4 = j.accumulate ? reinterpret_cast<float4*>(j.dw)[e4] : f4_zero();
#pragma unroll
      for (int i = 0; i < 16; ++i) t4 = f4_add(t4, red[i][vv]);
      reinterpret_cast<float4*>(j.dw)[e4] = t4;
      const AdamScalars sc = adam_scalars(lr, beta1, beta2, eps, t_sh);
      float4 p4 = reinterpret_cast<float4*>(j.param)[e4], m4 = reinterpret_cast<float4*>(j.m)[e4],
             v4 = reinterpret_cast<float4*>(j.v)[e4];
      adam_update(p4.x, m4.x, v4.x, t4.x, sc);
      adam_update(p4.y, m4.y, v4.y, t4.y, sc);
      adam_update(p4.z, m4.z, v4.z, t4.z, sc);
      adam_update(p4.w, m4.w, v4.w, t4.w, sc);
      reinterpret_cast<float4*>(j.param)[e4] = p4;
      reinterpret_cast<float4*>(j.m)[e4] = m4;
      reinterpret_cast<float4*>(j.v)[e4] = v4;
    }
    return;
  }
  // ---- the finalize block (same arithmetic and order as loss_finalize_kernel)
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  for (int i = tid; i < fin.n1; i += 256) { a += fin.p1[2 * i]; b += fin.p1[2 * i + 1]; }
  for (int i = tid; i < fin.n2; i += 256) { c += fin.p2[2 * i]; d += fin.p2[2 * i + 1]; }
  fred[0][tid] = a; fred[1][tid] = b; fred[2][tid] = c; fred[3][tid] = d;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
#pragma unroll
      for (int q = 0; q < 4; ++q) fred[q][tid] += fred[q][tid + off];
    }
    __syncthreads();
  }
  if (tid < 4) fin.hist[(int64_t)(*fin.pos) * 4 + tid] = fred[tid][0];
  __syncthreads();
  if (tid == 0) {
    while (atomicAdd(fin.arrive, 0) < nb1 + nb2) __builtin_amdgcn_s_sleep(8);
    *fin.arrive = 0;
    *fin.pos = (*fin.pos + 1) % fin.capacity;
    atomicAdd(fin.iter, 1);
  }
}

// any n_elem (not a multiple of 4): one thread per element
__global__ __launch_bounds__(256) void wgrad_reduce_scalar_kernel(const float* __restrict__ partials, int32_t n_part,
                                                                  int32_t n_elem, int32_t accumulate,
                                                                  float* __restrict__ dw) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_elem) return;
  float s = accumulate ? dw[e] : 0.f;
  for (int b = 0; b < n_part; ++b) s += partials[(int64_t)b * n_elem + e];
  dw[e] = s;
}

static inline void wgrad_geometry(int32_t n_sel, int* n_blocks, int* rows_per_block) {
  int nb = (n_sel + 127) / 128;
  if (nb > 512) nb = 512;
  if (nb < (n_sel + kWgradMaxRows - 1) / kWgradMaxRows) nb = (n_sel + kWgradMaxRows - 1) / kWgradMaxRows;
  if (nb < 1) nb = 1;
  int rpb = (n_sel + nb - 1) / nb;
  rpb = (rpb + 31) / 32 * 32;
  if (rpb < 32) rpb = 32;
  *n_blocks = (n_sel + rpb - 1) / rpb;
  if (*n_blocks < 1) *n_blocks = 1;
  *rows_per_block = rpb;
}

}  // namespace gd

static int rows_gemm_impl(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel, const float* w,
                          int32_t d_in, int32_t d_out, int32_t trans_w, const float* bias, int32_t relu_in,
                          const uint32_t* gate_bits, uint32_t* sign_out, float* out, int64_t ld_out,
                          float* save_in, void* stream, const float* in_alt = nullptr, const uint8_t* sel = nullptr,
                          gd::RowDots dots = gd::RowDots{nullptr, nullptr, nullptr, nullptr}) {
  using namespace gd;
  GD_REQUIRE(!sel || (in_alt && aligned16(in_alt) && in_alt != out && !gate_bits && !sign_out), GD_E_NULL,
             "gd_rows_gemm_select_f32: bad in_alt");
  GD_REQUIRE(in && w && out, GD_E_NULL, "gd_rows_gemm_f32: null pointer");
  GD_REQUIRE(n_sel >= 0 && d_in > 0 && d_out > 0 && ld_in >= d_in && ld_out >= d_out, GD_E_DIM,
             "gd_rows_gemm_f32: bad dims n_sel=%d d_in=%d d_out=%d", n_sel, d_in, d_out);
  GD_REQUIRE(in != out || d_in == d_out, GD_E_DIM, "gd_rows_gemm_f32: in-place needs d_in == d_out");
  GD_REQUIRE(!(gate_bits && sign_out), GD_E_DIM, "gd_rows_gemm_f32: gate and sign output are exclusive");
  if (n_sel == 0) return GD_OK;
  {   // the Del operator's widths at step size: weight-stationary register form (rows_gemm_ws.hip)
    const int rc = rows_gemm_ws_try(in, ld_in, idx, n_sel, w, d_in, d_out, trans_w, bias, relu_in, gate_bits, sign_out, out, ld_out,
                                    save_in, stream, in_alt, sel, dots.u1, dots.u2, dots.o1, dots.o2);
    if (rc != 1) return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  size_t lds = (size_t)d_in * 32 * (d_out / 32 == 3 ? 4 : d_out / 32) * sizeof(float);   // NT = 3 is padded to 4
  const int np = matrix_split();
  const bool mfma_ok = (d_out % 32 == 0) && d_out <= 128 && (d_in % 32 == 0) && lds <= 64 * 1024 && aligned16(in) &&
                       (ld_in % 4 == 0) && (!save_in || aligned16(save_in)) && aligned16(out) && (ld_out % 4 == 0) &&
                       (!bias || aligned16(bias));
  if (mfma_ok) {
    const int n_tiles = (n_sel + 31) / 32;
    int grid = (n_tiles + 7) / 8;
    // persistent blocks, 2 per CU by default (GD_ROWS_GEMM_GRID: A-B knob, read once)
    static const int grid_cap = [] { const char* e = getenv("GD_ROWS_GEMM_GRID"); return e && atoi(e) > 0 ? atoi(e) : 512; }();
    if (grid > grid_cap) grid = grid_cap;
    // fewer tiles than a wave per SIMD and CU: one block per compute unit, its tiles on different SIMDs (wave-major hand-out)
    static const bool spread_on = [] { const char* e = getenv("GD_ROWS_GEMM_SPREAD"); return !(e && atoi(e) == 0); }();
    if (spread_on && grid < ws_cu_count()) grid = n_tiles < ws_cu_count() ? n_tiles : ws_cu_count();
    // split arithmetic where its register-resident rows are instantiated (d_in = 64 / 128); other widths keep the fp32 instruction
    // (96 / 128 outputs with whole rows in registers; the 128 -> 64 product chunk-wise, which keeps 4 waves per SIMD)
    const int kc_split = (np && (d_in == 64 || d_in == 128) && d_out >= 96) ? d_in / 32 : 0;
    const bool narrow_split = np && !kc_split && d_out == 64 && d_in == 128;      // the 128 -> 64 product of the step
    if (kc_split || narrow_split) {
      lds = (size_t)3 * d_in * d_out * 2;                    // three bf16 images, no padding of NT = 3
      if (grid > grid_cap / 2 && lds > 80 * 1024) grid = grid_cap / 2;      // one block per CU fits
    }
    // tile queue (one 16-wave block per CU) where a block gets enough tiles to hand out: the step's N- and S-row products
    static const bool queue_on = [] { const char* e = getenv("GD_ROWS_GEMM_QUEUE"); return !(e && atoi(e) == 0); }();
    const bool use_queue = queue_on && !kc_split && !narrow_split && n_tiles >= 16 * 256;
    if (use_queue && grid > ws_cu_count()) grid = ws_cu_count();            // one 16-wave block per compute unit
#define GD_RG_KERNEL(NT, MODE, SELV, NPV, KCV)                                                                    \
  do {                                                                                                            \
    if (use_queue && !NPV) {                                                                                      \
      auto kq = rows_gemm_mfma_kernel<NT, MODE, SELV, 0, 0, true>;                                                \
      /* the tile counter is static LDS on top of the dynamic weight image (exactly 64 KB at 128 x 128) */        \
      static const hipError_t onceq = hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024 + 64); \
      GD_REQUIRE(onceq == hipSuccess, -(int)onceq, "gd_rows_gemm_f32: cannot raise the LDS limit of the tile-queue kernel"); \
      hipLaunchKernelGGL(kq, dim3(grid), dim3(1024), lds, s, in, ld_in, idx, n_sel, w, d_in, trans_w,             \
                         bias, relu_in, out, ld_out, save_in, gate_bits, sign_out, in_alt, sel, dots);            \
      break;                                                                                                      \
    }                                                                                                             \
    auto kern = rows_gemm_mfma_kernel<NT, MODE, SELV, NPV, KCV>;                                                  \
    if (NPV && lds > 64 * 1024) {                                                                                 \
      static const hipError_t once = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      GD_REQUIRE(once == hipSuccess, -(int)once, "gd_rows_gemm_f32: cannot raise the LDS limit");                \
    }                                                                                                             \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kGemmThreads), lds, s, in, ld_in, idx, n_sel, w, d_in, trans_w,     \
                       bias, relu_in, out, ld_out, save_in, gate_bits, sign_out, in_alt, sel, dots);              \
  } while (0)
#define GD_RG_LAUNCH(NT, MODE, SELV)                                                                              \
  do {                                                                                                            \
    if (kc_split == 4) GD_RG_KERNEL(NT, MODE, SELV, 6, 4);                                                        \
    else if (kc_split == 2) GD_RG_KERNEL(NT, MODE, SELV, 6, 2);                                                   \
    else if (narrow_split && NT == 2) GD_RG_KERNEL(2, MODE, SELV, 6, 0);                                          \
    else GD_RG_KERNEL(NT, MODE, SELV, 0, 0);                                                                      \
  } while (0)
#define GD_RG_CASE(NT)                                                                                            \
  do {                                                                                                            \
    if (gate_bits && dots.u1) GD_RG_LAUNCH(NT, 4, false);                                                         \
    else if (gate_bits) GD_RG_LAUNCH(NT, 2, false);                                                               \
    else if (sign_out) GD_RG_LAUNCH(NT, 1, false);                                                                \
    else if (dots.u1 && sel) GD_RG_LAUNCH(NT, 3, true);                                                           \
    else if (dots.u1) GD_RG_LAUNCH(NT, 3, false);                                                                 \
    else if (sel) GD_RG_LAUNCH(NT, 0, true);                                                                      \
    else GD_RG_LAUNCH(NT, 0, false);                                                                              \
  } while (0)
    switch (d_out / 32) {
      case 1: GD_RG_CASE(1); break;
      case 2: GD_RG_CASE(2); break;
      case 3: GD_RG_CASE(3); break;
      default: GD_RG_CASE(4); break;
    }
#undef GD_RG_KERNEL
#undef GD_RG_LAUNCH
#undef GD_RG_CASE
    return launched("rows_gemm_mfma");
  }
  GD_REQUIRE(!dots.u1, GD_E_DIM, "gd_rows_gemm_dots_f32: needs the MFMA path (d_in, d_out multiples of 32, d_out <= 128, "
             "weight <= 64 KB, 16-byte aligned rows)");
  GD_REQUIRE(d_in <= 1024, GD_E_DIM, "gd_rows_gemm_f32: fallback path needs d_in <= 1024 (got %d)", d_in);
  hipLaunchKernelGGL(rows_gemm_scalar_kernel, dim3((n_sel + 3) / 4), dim3(256), 0, s, in, ld_in, idx, n_sel, w, d_in,
                     d_out, trans_w, bias, relu_in, out, ld_out, save_in, gate_bits, sign_out, in_alt, sel);
  return launched("rows_gemm_scalar");
}

extern "C" int gd_rows_gemm_f32(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel, const float* w,
                                int32_t d_in, int32_t d_out, int32_t trans_w, const float* bias, int32_t relu_in,
                                float* out, int64_t ld_out, float* save_in, void* stream) {
  return rows_gemm_impl(in, ld_in, idx, n_sel, w, d_in, d_out, trans_w, bias, relu_in, nullptr, nullptr, out, ld_out,
                        save_in, stream);
}

extern "C" int gd_rows_gemm_select_f32(const float* in, const float* in_alt, const uint8_t* sel, int64_t ld_in,
                                       const int32_t* idx, int32_t n_sel, const float* w, int32_t d_in, int32_t d_out,
                                       int32_t trans_w, const float* bias, int32_t relu_in, float* out, int64_t ld_out,
                                       void* stream) {
  GD_REQUIRE(sel, GD_E_NULL, "gd_rows_gemm_select_f32: null selector");
  return rows_gemm_impl(in, ld_in, idx, n_sel, w, d_in, d_out, trans_w, bias, relu_in, nullptr, nullptr, out, ld_out,
                        nullptr, stream, in_alt, sel);
}

extern "C" int gd_rows_gemm_dots_f32(const float* in, const float* in_alt, const uint8_t* sel, int64_t ld_in,
                                     const float* w, int32_t d_in, int32_t d_out, int32_t trans_w, const float* bias,
                                     int32_t relu_in, float* out, int64_t ld_out, const int32_t* idx, int32_t n_rows,
                                     const float* u1, const float* u2, float* o1, float* o2, void* stream) {
  using namespace gd;
  GD_REQUIRE(u1 && u2 && o1 && o2 && aligned16(u1) && aligned16(u2), GD_E_NULL,
             "gd_rows_gemm_dots_f32: u1 / u2 (16-byte aligned) and o1 / o2 are required");
  GD_REQUIRE(in != out, GD_E_DIM, "gd_rows_gemm_dots_f32: in and out must not alias");
  return rows_gemm_impl(in, ld_in, idx, n_rows, w, d_in, d_out, trans_w, bias, relu_in, nullptr, nullptr, out, ld_out,
                        nullptr, stream, sel ? in_alt : nullptr, sel, RowDots{u1, u2, o1, o2});
}

extern "C" int gd_rows_gemm_signs_f32(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel,
                                      const float* w, int32_t d_in, int32_t d_out, int32_t trans_w, const float* bias,
                                      int32_t relu_in, float* out, int64_t ld_out, float* save_in,
                                      uint32_t* sign_bits, void* stream) {
  GD_REQUIRE(sign_bits, GD_E_NULL, "gd_rows_gemm_signs_f32: null sign_bits");
  return rows_gemm_impl(in, ld_in, idx, n_sel, w, d_in, d_out, trans_w, bias, relu_in, nullptr, sign_bits, out, ld_out,
                        save_in, stream);
}

extern "C" int gd_rows_gemm_gated_f32(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel,
                                      const float* w, int32_t d_in, int32_t d_out, int32_t trans_w,
                                      const uint32_t* gate_bits, float* out, int64_t ld_out, void* stream) {
  GD_REQUIRE(gate_bits, GD_E_NULL, "gd_rows_gemm_gated_f32: null gate");
  return rows_gemm_impl(in, ld_in, idx, n_sel, w, d_in, d_out, trans_w, nullptr, 0, gate_bits, nullptr, out, ld_out,
                        nullptr, stream);
}

extern "C" int gd_rows_gemm_gated_rank1_f32(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel,
                                            const float* w, int32_t d_in, int32_t d_out, int32_t trans_w,
                                            const uint32_t* gate_bits, const float* row_a, const float* col_p,
                                            const float* row_b, const float* col_q, float* out, int64_t ld_out,
                                            void* stream) {
  GD_REQUIRE(gate_bits && row_a && col_p && row_b && col_q, GD_E_NULL, "gd_rows_gemm_gated_rank1_f32: null pointer");
  GD_REQUIRE(gd::aligned16(col_p) && gd::aligned16(col_q), GD_E_ALIGN, "gd_rows_gemm_gated_rank1_f32: unaligned column vectors");
  return rows_gemm_impl(in, ld_in, idx, n_sel, w, d_in, d_out, trans_w, nullptr, 0, gate_bits, nullptr, out, ld_out,
                        nullptr, stream, nullptr, nullptr,
                        gd::RowDots{col_p, col_q, const_cast<float*>(row_a), const_cast<float*>(row_b)});
}

extern "C" int64_t gd_rows_gemm_wgrad_workspace(int32_t n_sel, int32_t d_a, int32_t d_b) {
  int nb, rpb;
  gd::wgrad_geometry(n_sel, &nb, &rpb);
  return (int64_t)nb * d_a * d_b;
}

namespace gd {
struct AdamArgs { float* param; float* m; float* v; const int32_t* iter; double lr, beta1, beta2, eps; };
}

static int wgrad_impl(const gd::WgradLoss* loss, const float* a, int64_t ld_a, const int32_t* a_idx, const float* g,
                      int64_t ld_g, const int32_t* g_idx, const float* relu_mask, const float* g_add, int32_t n_sel, int32_t d_a, int32_t d_b, float* dw,
                      int32_t accumulate, float* partials, const gd::AdamArgs* adam, void* stream, bool reduce_only = false) {
  using namespace gd;
  GD_REQUIRE(partials, GD_E_NULL, "gd_rows_gemm_wgrad_f32: null partials");
  GD_REQUIRE(dw || !adam, GD_E_NULL, "gd_rows_gemm_wgrad_f32: an optimizer step needs dw");
  GD_REQUIRE(n_sel == 0 || reduce_only || (a && g), GD_E_NULL, "gd_rows_gemm_wgrad_f32: null input");
  GD_REQUIRE(n_sel >= 0 && d_a > 0 && d_b > 0 && ld_a >= d_a && ld_g >= d_b, GD_E_DIM,
             "gd_rows_gemm_wgrad_f32: bad dims");
  hipStream_t s = (hipStream_t)stream;
  const int n_elem = d_a * d_b;
  int nb = 0, rpb = 8;
  if (n_sel > 0 && reduce_only) wgrad_geometry(n_sel, &nb, &rpb);
  if (n_sel > 0 && !reduce_only) {
    wgrad_geometry(n_sel, &nb, &rpb);
    const int ta = d_a / 32, tb = d_b / 32;
    const bool mfma_ok = (d_a % 32 == 0) && (d_b % 32 == 0) && ta <= 4 && tb <= 4 && ta != 3 && tb != 3 &&
                         aligned16(a) && aligned16(g) && ld_a % 4 == 0 && ld_g % 4 == 0 &&
                         (!relu_mask || aligned16(relu_mask)) && (!g_add || aligned16(g_add));
    if (mfma_ok) {
    const WgradLoss no_loss{nullptr, nullptr, nullptr, nullptr, nullptr};
    // 128 x 128: 8 waves share the LDS tiles (2 output tiles each, 4 waves / SIMD at 2 blocks per CU) - measured
    // +3 % on the whole step over 4 waves with 4 tiles each; the smaller shapes keep 4 waves
    if (ta == 4 && tb == 4) {
      if (loss)
        hipLaunchKernelGGL((rows_wgrad_mfma_kernel<4, 4, true, 8>), dim3(nb), dim3(512), 0, s, a, ld_a, a_idx, g, ld_g,
                           g_idx, relu_mask, g_add, n_sel, rpb, partials, *loss);
      else
        hipLaunchKernelGGL((rows_wgrad_mfma_kernel<4, 4, false, 8>), dim3(nb), dim3(512), 0, s, a, ld_a, a_idx, g, ld_g,
                           g_idx, relu_mask, g_add, n_sel, rpb, partials, no_loss);
    } else
#define GD_WG_CASE(TA, TB)                                                                                        \
  do {                                                                                                            \
    if (loss)                                                                                                     \
      hipLaunchKernelGGL((rows_wgrad_mfma_kernel<TA, TB, true>), dim3(nb), dim3(256), 0, s, a, ld_a, a_idx, g,     \
                         ld_g, g_idx, relu_mask, g_add, n_sel, rpb, partials, *loss);                             \
    else                                                                                                          \
      hipLaunchKernelGGL((rows_wgrad_mfma_kernel<TA, TB, false>), dim3(nb), dim3(256), 0, s, a, ld_a, a_idx, g,    \
                         ld_g, g_idx, relu_mask, g_add, n_sel, rpb, partials, no_loss);                           \
  } while (0)
      switch (ta * 8 + tb) {
        case 1 * 8 + 1: GD_WG_CASE(1, 1); break;
        case 1 * 8 + 2: GD_WG_CASE(1, 2); break;
        case 1 * 8 + 4: GD_WG_CASE(1, 4); break;
        case 2 * 8 + 1: GD_WG_CASE(2, 1); break;
        case 2 * 8 + 2: GD_WG_CASE(2, 2); break;
        case 2 * 8 + 4: GD_WG_CASE(2, 4); break;
        case 4 * 8 + 1: GD_WG_CASE(4, 1); break;
        case 4 * 8 + 2: GD_WG_CASE(4, 2); break;
        default: GD_WG_CASE(4, 4); break;
      }
#undef GD_WG_CASE
    } else {
      GD_REQUIRE(!loss, GD_E_DIM, "gd_rows_gemm_wgrad_loss_f32: the fused loss form needs widths in {32,64,128}");
      hipLaunchKernelGGL(rows_wgrad_scalar_kernel, dim3(nb), dim3(256), 0, s, a, ld_a, a_idx, g, ld_g, g_idx,
                         relu_mask, g_add, n_sel, d_a, d_b, rpb, partials);
    }
    int rc = launched("rows_wgrad");
    if (rc) return rc;
  }
  if (!dw) return GD_OK;       // partial products only: the caller reduces them (gd_rows_gemm_wgrad_reduce_f32)
  const bool vec = n_elem % 4 == 0 && aligned16(dw) && aligned16(partials);
  if (adam && vec) {
    hipLaunchKernelGGL(wgrad_reduce_adam_kernel, dim3((n_elem / 4 + 15) / 16), dim3(256), 0, s, partials, nb, n_elem,
                       accumulate, dw, adam->param, adam->m, adam->v, adam->iter, adam->lr, adam->beta1, adam->beta2,
                       adam->eps);
    return launched("wgrad_reduce_adam");
  }
  if (vec)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n_elem / 4 + 15) / 16), dim3(256), 0, s, partials, nb, n_elem,
                       accumulate, dw);
  else
    hipLaunchKernelGGL(wgrad_reduce_scalar_kernel, dim3((n_elem + 255) / 256), dim3(256), 0, s, partials, nb, n_elem,
                       accumulate, dw);
  int rc = launched("wgrad_reduce");
  if (rc || !adam) return rc;
  return gd_adam_at_f32(adam->param, dw, adam->m, adam->v, adam->iter, n_elem, adam->lr, adam->beta1, adam->beta2,
                        adam->eps, stream);
}

extern "C" int gd_rows_gemm_wgrad_f32(const float* a, int64_t ld_a, const int32_t* a_idx, const float* g, int64_t ld_g,
                                      const int32_t* g_idx, const float* relu_mask, const float* g_add,
                                      int32_t n_sel, int32_t d_a, int32_t d_b, float* dw, int32_t accumulate,
                                      float* partials, void* stream) {
  return wgrad_impl(nullptr, a, ld_a, a_idx, g, ld_g, g_idx, relu_mask, g_add, n_sel, d_a, d_b, dw, accumulate,
                    partials, nullptr, stream);
}

extern "C" int gd_rows_gemm_wgrad_adam_f32(const float* a, int64_t ld_a, const int32_t* a_idx, const float* g,
                                           int64_t ld_g, const int32_t* g_idx, const float* relu_mask,
                                           const float* g_add, int32_t n_sel, int32_t d_a, int32_t d_b, float* dw,
                                           int32_t accumulate, float* partials,
                                           float* param, float* exp_avg, float* exp_avg_sq, const int32_t* iter,
                                           double lr, double beta1, double beta2, double eps, void* stream) {
  GD_REQUIRE(param && exp_avg && exp_avg_sq && iter, GD_E_NULL, "gd_rows_gemm_wgrad_adam_f32: null optimizer state");
  const gd::AdamArgs adam{param, exp_avg, exp_avg_sq, iter, lr, beta1, beta2, eps};
  return wgrad_impl(nullptr, a, ld_a, a_idx, g, ld_g, g_idx, relu_mask, g_add, n_sel, d_a, d_b, dw, accumulate,
                    partials, &adam, stream);
}

extern "C" int gd_rows_gemm_wgrad_reduce_f32(const float* partials, int32_t n_sel, int32_t d_a, int32_t d_b, float* dw,
                                             int32_t accumulate, float* param, float* exp_avg, float* exp_avg_sq,
                                             const int32_t* iter, double lr, double beta1, double beta2, double eps,
                                             void* stream) {
  GD_REQUIRE(partials && dw, GD_E_NULL, "gd_rows_gemm_wgrad_reduce_f32: null pointer");
  GD_REQUIRE(!param || (exp_avg && exp_avg_sq && iter), GD_E_NULL, "gd_rows_gemm_wgrad_reduce_f32: null optimizer state");
  GD_REQUIRE(n_sel >= 0 && d_a > 0 && d_b > 0, GD_E_DIM, "gd_rows_gemm_wgrad_reduce_f32: bad dims");
  const gd::AdamArgs adam{param, exp_avg, exp_avg_sq, iter, lr, beta1, beta2, eps};
  // n_sel = 0 rows: no partials exist, dW (+)= 0 (wgrad_impl with nothing to launch but the reduction)
  return wgrad_impl(nullptr, nullptr, d_a, nullptr, nullptr, d_b, nullptr, nullptr, nullptr, n_sel, d_a, d_b, dw, accumulate,
                    const_cast<float*>(partials), param ? &adam : nullptr, stream, true);
}

static int step_tail_impl(const char* name, const float* partials1, int32_t nb1, int32_t d1, int32_t accumulate1, float* dw1, float* param1,
                          float* exp_avg1, float* exp_avg_sq1, const float* partials2, int32_t nb2, int32_t d2, int32_t accumulate2,
                          float* dw2, float* param2, float* exp_avg2, float* exp_avg_sq2, double lr, double beta1, double beta2,
                          double eps, const float* loss_partials1, int32_t n1, const float* loss_partials2, int32_t n2, float* hist,
                          int32_t capacity, int32_t* pos, int32_t* iter, int32_t* arrive, void* stream) {
  using namespace gd;
  GD_REQUIRE(partials1 && dw1 && param1 && exp_avg1 && exp_avg_sq1 && partials2 && dw2 && param2 && exp_avg2 && exp_avg_sq2, GD_E_NULL,
             "%s: null weight-gradient / optimizer pointer", name);
  GD_REQUIRE(hist && pos && iter && arrive && capacity > 0 && (n1 == 0 || loss_partials1) && (n2 == 0 || loss_partials2), GD_E_NULL,
             "%s: null bookkeeping pointer", name);
  GD_REQUIRE(nb1 > 0 && nb2 > 0 && d1 > 0 && d2 > 0 && (d1 * d1) % 4 == 0 && (d2 * d2) % 4 == 0, GD_E_DIM,
             "%s: both Del weights need rows and widths that are multiples of 2", name);
  GD_REQUIRE(aligned16(partials1) && aligned16(dw1) && aligned16(partials2) && aligned16(dw2), GD_E_ALIGN, "%s: unaligned", name);
  const TailJob j1{partials1, nb1, d1 * d1, accumulate1, dw1, param1, exp_avg1, exp_avg_sq1};
  const TailJob j2{partials2, nb2, d2 * d2, accumulate2, dw2, param2, exp_avg2, exp_avg_sq2};
  const TailFin fin{loss_partials1, n1, loss_partials2, n2, hist, capacity, pos, iter, arrive};
  const int grid = (j1.n_elem / 4 + 15) / 16 + (j2.n_elem / 4 + 15) / 16 + 1;
  GD_REQUIRE(grid <= 1024, GD_E_DIM, "%s: Del weights too large for the co-resident grid (%d blocks)", name, grid);
  hipLaunchKernelGGL(step_tail_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, j1, j2, fin, lr, beta1, beta2, eps);
  return launched("step_tail");
}

extern "C" int gd_step_tail_f32(const float* partials1, int32_t n_sel1, int32_t d1, int32_t accumulate1, float* dw1, float* param1,
                                float* exp_avg1, float* exp_avg_sq1, const float* partials2, int32_t n_sel2, int32_t d2,
                                int32_t accumulate2, float* dw2, float* param2, float* exp_avg2, float* exp_avg_sq2, double lr,
                                double beta1, double beta2, double eps, const float* loss_partials1, int32_t n1,
                                const float* loss_partials2, int32_t n2, float* hist, int32_t capacity, int32_t* pos, int32_t* iter,
                                int32_t* arrive, void* stream) {
  int nb1 = 0, nb2 = 0, rpb;
  if (n_sel1 > 0) gd::wgrad_geometry(n_sel1, &nb1, &rpb);
  if (n_sel2 > 0) gd::wgrad_geometry(n_sel2, &nb2, &rpb);
  return step_tail_impl("gd_step_tail_f32", partials1, nb1, d1, accumulate1, dw1, param1, exp_avg1, exp_avg_sq1, partials2, nb2, d2, accumulate2,
                        dw2, param2, exp_avg2, exp_avg_sq2, lr, beta1, beta2, eps, loss_partials1, n1, loss_partials2, n2, hist, capacity,
                        pos, iter, arrive, stream);
}

extern "C" int gd_step_tail_parts_f32(const float* partials1, int32_t n_part1, int32_t d1, int32_t accumulate1, float* dw1, float* param1,
                                      float* exp_avg1, float* exp_avg_sq1, const float* partials2, int32_t n_part2, int32_t d2,
                                      int32_t accumulate2, float* dw2, float* param2, float* exp_avg2, float* exp_avg_sq2, double lr,
                                      double beta1, double beta2, double eps, const float* loss_partials1, int32_t n1,
                                      const float* loss_partials2, int32_t n2, float* hist, int32_t capacity, int32_t* pos,
                                      int32_t* iter, int32_t* arrive, void* stream) {
  return step_tail_impl("gd_step_tail_parts_f32", partials1, n_part1, d1, accumulate1, dw1, param1, exp_avg1, exp_avg_sq1, partials2, n_part2, d2,
                        accumulate2, dw2, param2, exp_avg2, exp_avg_sq2, lr, beta1, beta2, eps, loss_partials1, n1, loss_partials2, n2,
                        hist, capacity, pos, iter, arrive, stream);
}

extern "C" int32_t gd_rows_gemm_wgrad_blocks(int32_t n_sel) {
  int nb, rpb;
  gd::wgrad_geometry(n_sel, &nb, &rpb);
  return n_sel > 0 ? nb : 0;
}

extern "C" int gd_rows_gemm_wgrad_loss_f32(const float* a, int64_t ld_a, const int32_t* a_idx, const float* z,
                                           int64_t ld_z, const int32_t* z_idx, const int32_t* loss_slot,
                                           const float* tm, const float* coef, const float* cnt_signed,
                                           const float* g_add, int32_t n_sel, int32_t d_a, int32_t d_b, float* dw,
                                           int32_t accumulate, float* partials, float* loss_partials, float* param,
                                           float* exp_avg, float* exp_avg_sq, const int32_t* iter, double lr,
                                           double beta1, double beta2, double eps, void* stream) {
  GD_REQUIRE(n_sel == 0 || (loss_slot && tm && coef && cnt_signed), GD_E_NULL, "gd_rows_gemm_wgrad_loss_f32: null loss terms");
  GD_REQUIRE(loss_partials, GD_E_NULL, "gd_rows_gemm_wgrad_loss_f32: null loss_partials");
  GD_REQUIRE(!param || (exp_avg && exp_avg_sq && iter), GD_E_NULL, "gd_rows_gemm_wgrad_loss_f32: null optimizer state");
  GD_REQUIRE(n_sel == 0 || gd::aligned16(tm), GD_E_ALIGN, "gd_rows_gemm_wgrad_loss_f32: unaligned targets");
  const gd::WgradLoss loss{loss_slot, tm, coef, cnt_signed, loss_partials};
  const gd::AdamArgs adam{param, exp_avg, exp_avg_sq, iter, lr, beta1, beta2, eps};
  return wgrad_impl(&loss, a, ld_a, a_idx, z, ld_z, z_idx, nullptr, g_add, n_sel, d_a, d_b, dw, accumulate, partials,
                    param ? &adam : nullptr, stream);
}

// out[idx[s], :] = src[idx[s], :] where bit (s, c) of the packed sign pattern is set, else 0: the ReLU backward on a
// row subset from the [z > 0] bits gd_rows_gemm_signs_f32 recorded (one launch; the tensor-op form took eight).
namespace gd {
__global__ __launch_bounds__(256) void gate_rows_kernel(const float* __restrict__ src, int64_t ld_src, const int32_t* __restrict__ idx,
                                                        int32_t n_sel, const uint32_t* __restrict__ bits, int32_t d4, int32_t n_words,
                                                        float* __restrict__ out, int64_t ld_out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)n_sel * d4) return;
  const int s = (int)(e / d4), c4 = (int)(e % d4);
  const int64_t row = idx ? idx[s] : s;
  const uint32_t m = bits[(int64_t)s * n_words + (c4 >> 3)] >> (4 * (c4 & 7));
  float4 v = reinterpret_cast<const float4*>(src + row * ld_src)[c4];
  v.x = (m & 1u) ? v.x : 0.f; v.y = (m & 2u) ? v.y : 0.f; v.z = (m & 4u) ? v.z : 0.f; v.w = (m & 8u) ? v.w : 0.f;
  reinterpret_cast<float4*>(out + row * ld_out)[c4] = v;
}
}  // namespace gd

extern "C" int gd_gate_rows_f32(const float* src, int64_t ld_src, const int32_t* idx, int32_t n_sel, const uint32_t* gate_bits,
                                int32_t d, float* out, int64_t ld_out, void* stream) {
  using namespace gd;
  GD_REQUIRE(src && gate_bits && out, GD_E_NULL, "gd_gate_rows_f32: null pointer");
  GD_REQUIRE(n_sel >= 0 && d > 0 && d % 4 == 0 && ld_src >= d && ld_out >= d && ld_src % 4 == 0 && ld_out % 4 == 0, GD_E_DIM,
             "gd_gate_rows_f32: d and the row pitches must be multiples of 4 (d=%d)", d);
  GD_REQUIRE(aligned16(src) && aligned16(out), GD_E_ALIGN, "gd_gate_rows_f32: unaligned pointer");
  if (n_sel == 0) return GD_OK;
  const int64_t total = (int64_t)n_sel * (d / 4);
  hipLaunchKernelGGL(gate_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src, idx,
                     n_sel, gate_bits, d / 4, (d + 31) / 32, out, ld_out);
  return launched("gate_rows");
}
