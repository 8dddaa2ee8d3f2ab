// Aggregate-then-transform in ONE kernel:   y[r,:] = ( sum_k val[k] x[col[k],:] ) @ W (+ bias)   for r in rows
//
// = a GCN / GIN layer whose input is no wider than its output (framework/models/gcn.py:11-24, gin.py:26-34;
// by linearity A (x W^T) = (A x) W^T), and the layer-2 input gradient (A^T dp2)[S1] W2 with the ReLU gate.
// Run separately, the aggregation is bound by the gather path (the matrix cores idle for ~100 us) and the
// transform by the matrix cores (the memory system idles for ~90 us), and running the two KERNELS side by side
// only makes them time-share the CUs' wave slots.  Here every wave alternates between the
// two phases on its own 16-row tile, so while some waves of a CU wait for gathered rows others feed the MFMA
// pipe; the aggregated tile never leaves the CU (registers -> 8 KB of LDS -> MFMA operand registers).
//
//   phase A  the wave walks its tile's rows; a row of d_in floats is covered by LPR = d_in/4 lanes x float4, so
//            G = 64/LPR neighbours are gathered per load instruction.  The loads of the NEXT batch (up to U
//            instructions, possibly of the next row) are issued before the current batch is consumed: few waves
//            per CU (the 64 KB weight image limits them to NW) but 2U loads in flight per wave, which the
//            gather probe shows is enough (tools/experiments/gather_probe.hip: 8 waves/CU x 8 loads reach the
//            rate of 32 waves/CU).  Row sums are reduced over the lane groups and written to the wave's tile.
//   phase B  D^T[feat][sample] = W[feat][k] . X^T[k][sample] with v_mfma_f32_16x16x4_f32: lane (n = l % 16,
//            q = l / 16) holds sample n's features k in [KQ q, KQ q + KQ) (8 or 4 ds_read_b128 from the tile,
//            conflict-free with the +4 pitch) and, per output tile, the weight image pre-permuted so that its
//            operand is one ds_read_b128 per 4 MFMAs with consecutive lanes on consecutive 16 bytes.  Both
//            operands use the same k permutation (k = KQ q + 4 j + c at step (j, c)), which a dot product does
//            not see.  The lane ends with 4 consecutive output features of its sample: one 16-byte store.
//
// Rows are walked in segments of 64 edges by ONE wave, so very heavy rows would serialise: the caller
// pre-aggregates rows above a cap (256 edges) with the balanced SpMM into extra rows of x and gives them a
// single edge (gnndelete_amd/graph.py: CappedCSR).
#include "common.h"

namespace gd {

using v4f = __attribute__((ext_vector_type(4))) float;

template <int DIN, int DOUT, int NW, bool GATE>
__global__ __launch_bounds__(NW * 64) void agg_gemm_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rows, int32_t n_rows,
    const float* __restrict__ w, int32_t w_out_in, const float* __restrict__ bias,
    const uint32_t* __restrict__ gate_bits, float* __restrict__ y, int64_t ldy, int32_t nnz) {
  constexpr int LPR = DIN / 4;             // lanes per gathered row
  constexpr int G = kWave / LPR;           // rows per gather instruction
  constexpr int U = 4;                     // gather instructions per batch
  constexpr int PITCH = DIN + 4;           // tile row pitch in floats
  constexpr int KQ = DIN / 4;              // features per lane group q in phase B
  constexpr int J = KQ / 4;                // float4 operand fragments per lane
  constexpr int TOUT = DOUT / 16;          // 16-wide output tiles
  constexpr int kXcd = 8;
  __shared__ __attribute__((aligned(16))) float wimg[DIN * DOUT];
  __shared__ __attribute__((aligned(16))) float tiles[NW][16 * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- weight image: wimg[((t J + j) 64 + l) 4 + c] = W[feat = 16 t + l % 16][k = KQ (l / 16) + 4 j + c]
  for (int e = tid; e < DIN * DOUT; e += NW * 64) {
    const int c = e & 3, l = (e >> 2) & 63, tj = e >> 8;
    const int j = tj % J, t = tj / J;
    const int feat = 16 * t + (l & 15), k = KQ * (l >> 4) + 4 * j + c;
    wimg[e] = w_out_in ? w[feat * DIN + k] : w[k * DOUT + feat];
  }
  __syncthreads();

  float* tile = tiles[wave];
  const int g = lane / LPR, li = lane % LPR;
  const uint32_t lo = 16u * (uint32_t)li;
  const uint32_t pitch_b = (uint32_t)ldx * 4u;
  const char* xb = reinterpret_cast<const char*>(x);
  const int n_tiles = (n_rows + 15) >> 4;
  const int xcd = blockIdx.x % kXcd;
  const int per = (n_tiles + kXcd - 1) / kXcd;
  const int t_first = xcd * per, t_last = min(n_tiles, t_first + per);
  // static, XCD-contiguous tile assignment: wave w of the XCD takes tiles first + w, first + w + W, ...
  // (one shared counter per XCD handing out tiles in order measured 35 % slower: 320 waves on one atomic)
  const int stride = (gridDim.x / kXcd) * NW;                   // waves per XCD
  const int wx = (blockIdx.x / kXcd) * NW + wave;

  // (visiting each XCD's tiles heaviest-first instead of in order measured 6 % slower: the order is the locality)
  for (int tl = t_first + wx; tl < t_last; tl += stride) {
    // ================= phase A: aggregate the tile's 16 rows into LDS
    // lane i < 16 holds row i's id and edge range (rows past the end are empty)
    const int slot = 16 * tl + (lane & 15);
    const bool live = slot < n_rows;
    const int rid = live ? (rows ? rows[slot] : slot) : 0;
    int rs = 0, re = 0;
    if (live) { rs = rowptr[rid]; re = rowptr[rid + 1]; }

    // Issue state (scalar): row being issued, its trip offset, its edge range.  EVERY batch issues exactly
    // U gathers and one (col, val) prefetch, unconditionally (padded trips re-read a cached row with weight 0,
    // index loads are clamped): with a fixed number of loads per batch the s_waitcnt in front of a batch's
    // consumer can leave the NEXT batch's loads in flight - with conditional loads the compiler has to wait
    // for everything and the pipeline collapses to one batch.
    // A row is walked in segments of at most 64 edges (what one (col, val) register pair holds); `rem` = edges
    // of the row behind the current segment.
    int ir = 0, it0 = 0;
    auto row_seg = [&](int r, int& s0, int& n0, int& rem0) {       // first segment of tile row r (clamped)
      const int rr = min(r, 15);
      s0 = __builtin_amdgcn_readlane(rs, rr);
      const int total = r < 16 ? __builtin_amdgcn_readlane(re, rr) - s0 : 0;
      n0 = min(total, 64);
      rem0 = total - n0;
    };
    auto next_seg = [&](int s0, int n0, int rem0, int r, int& s1, int& n1, int& rem1) {   // segment after (s0, n0, rem0)
      if (rem0 > 0) {
        s1 = s0 + n0;
        n1 = min(rem0, 64);
        rem1 = rem0 - n1;
      } else {
        row_seg(r + 1, s1, n1, rem1);
      }
    };
    int s_cur, cnt_cur, rem_cur, s_nxt, cnt_nxt, rem_nxt;
    row_seg(0, s_cur, cnt_cur, rem_cur);
    next_seg(s_cur, cnt_cur, rem_cur, 0, s_nxt, cnt_nxt, rem_nxt);
    int c_cur = col[min(s_cur + lane, nnz - 1)];
    float w_cur = val[min(s_cur + lane, nnz - 1)];
    int c_nxt = col[min(s_nxt + lane, nnz - 1)];
    float w_nxt = val[min(s_nxt + lane, nnz - 1)];
    struct Batch { float4 xv[U]; float wj[U]; int row; bool last; };
    auto issue = [&](Batch& b) {
      const int trips = (cnt_cur + G - 1) / G;
      b.row = ir;
      const int last4 = max(4 * cnt_cur - 4, 0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = (it0 + u) * G + g;                       // edge of the segment this lane group takes
        const int j4 = min(4 * j, last4);
        const int cs = __builtin_amdgcn_ds_bpermute(j4, c_cur);
        const float wv = __int_as_float(__builtin_amdgcn_ds_bpermute(j4, __float_as_int(w_cur)));
        b.wj[u] = j < cnt_cur ? wv : 0.f;
        b.xv[u] = *reinterpret_cast<const float4*>(xb + (__umul24((uint32_t)cs, pitch_b) + lo));
      }
      it0 += U;
      const bool seg_done = it0 >= trips;
      b.last = seg_done && rem_cur == 0;
      // advance to the next segment when this one is done (scalar work), and always refresh the prefetch of
      // the segment after the current one
      if (seg_done) {
        if (rem_cur == 0) ir += 1;
        it0 = 0;
        s_cur = s_nxt;
        cnt_cur = cnt_nxt;
        rem_cur = rem_nxt;
        c_cur = c_nxt;
        w_cur = w_nxt;
        next_seg(s_cur, cnt_cur, rem_cur, ir, s_nxt, cnt_nxt, rem_nxt);
      }
      c_nxt = col[min(s_nxt + lane, nnz - 1)];
      w_nxt = val[min(s_nxt + lane, nnz - 1)];
    };
    float4 acc = f4_zero();
    auto consume = [&](const Batch& b) {
#pragma unroll
      for (int u = 0; u < U; ++u) acc = f4_fma(b.wj[u], b.xv[u], acc);
      if (b.last) {
        acc = f4_group_sum<LPR>(acc);
        if (g == 0) *reinterpret_cast<float4*>(tile + b.row * PITCH + 4 * li) = acc;
        acc = f4_zero();
      }
    };
    // three batches rotate: two in flight while the oldest is consumed (every path is issue -> consume in
    // straight-line code so that the waits count exactly)
    Batch b0, b1, b2;
    issue(b0);
    if (ir >= 16) {
      consume(b0);
    } else {
      issue(b1);
      while (true) {
        if (ir >= 16) { consume(b0); consume(b1); break; }
        issue(b2);
        consume(b0);
        if (ir >= 16) { consume(b1); consume(b2); break; }
        issue(b0);
        consume(b1);
        if (ir >= 16) { consume(b2); consume(b0); break; }
        issue(b1);
        consume(b2);
      }
    }

    // ================= phase B: tile (16 x DIN) @ W -> 16 x DOUT
    const int n = lane & 15, q = lane >> 4;
    float4 xa[J];
#pragma unroll
    for (int j = 0; j < J; ++j) xa[j] = *reinterpret_cast<const float4*>(tile + n * PITCH + KQ * q + 4 * j);
    v4f dacc[TOUT];
#pragma unroll
    for (int t = 0; t < TOUT; ++t) {
      if (bias) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + 16 * t + 4 * q);
        dacc[t] = v4f{bv.x, bv.y, bv.z, bv.w};
      } else {
        dacc[t] = v4f{0.f, 0.f, 0.f, 0.f};
      }
    }
    const float4* wi = reinterpret_cast<const float4*>(wimg) + lane;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      // the weight fragments of ONE k group at a time (the compiler barrier keeps the scheduler from hoisting all
      // J x TOUT ds_reads - 256 registers - above the first MFMA)
      float4 wf[TOUT];
#pragma unroll
      for (int t = 0; t < TOUT; ++t) wf[t] = wi[(t * J + j) * 64];
#pragma unroll
      for (int t = 0; t < TOUT; ++t) dacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t].x, xa[j].x, dacc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TOUT; ++t) dacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t].y, xa[j].y, dacc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TOUT; ++t) dacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t].z, xa[j].z, dacc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TOUT; ++t) dacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t].w, xa[j].w, dacc[t], 0, 0, 0);
      asm volatile("" ::: "memory");
    }
    // lane (n, q) holds features 16 t + 4 q .. + 3 of sample n
    const int oslot = 16 * tl + n;
    if (oslot < n_rows) {
      float* yr = y + (int64_t)rid * ldy + 4 * q;      // rid: this lane's sample n = lane % 16

#pragma unroll
      for (int t = 0; t < TOUT; ++t) {
        float4 o = make_float4(dacc[t][0], dacc[t][1], dacc[t][2], dacc[t][3]);
        if (GATE) {
          const int f = 16 * t + 4 * q;
          const uint32_t bits = gate_bits[(int64_t)oslot * (DOUT / 32) + (f >> 5)] >> (f & 31);
          o.x = (bits & 1u) ? o.x : 0.f;
          o.y = (bits & 2u) ? o.y : 0.f;
          o.z = (bits & 4u) ? o.z : 0.f;
          o.w = (bits & 8u) ? o.w : 0.f;
        }
        *reinterpret_cast<float4*>(yr + 16 * t) = o;
      }
    }
  }
}

}  // namespace gd

extern "C" int gd_agg_gemm_f32(const int32_t* rowptr, const int32_t* col, const float* val, const float* x, int64_t ldx,
                               int32_t x_rows, const int32_t* rows, int32_t n_rows, const float* w, int32_t d_in,
                               int32_t d_out, int32_t w_out_in, const float* bias, const uint32_t* gate_bits, float* y,
                               int64_t ldy, int32_t nnz, void* stream) {
  using namespace gd;
  GD_REQUIRE(rowptr && col && val && x && w && y, GD_E_NULL, "gd_agg_gemm_f32: null pointer");
  GD_REQUIRE(nnz > 0, GD_E_DIM, "gd_agg_gemm_f32: nnz must be the length of col (> 0)");
  GD_REQUIRE((d_in == 64 || d_in == 128) && (d_out == 64 || d_out == 128), GD_E_DIM,
             "gd_agg_gemm_f32: d_in=%d d_out=%d must each be 64 or 128", d_in, d_out);
  GD_REQUIRE(n_rows >= 0 && ldx >= d_in && ldy >= d_out && ldx % 4 == 0 && ldy % 4 == 0, GD_E_DIM,
             "gd_agg_gemm_f32: bad leading dimensions");
  GD_REQUIRE(aligned16(x) && aligned16(y) && (!bias || aligned16(bias)), GD_E_ALIGN, "gd_agg_gemm_f32: unaligned pointer");
  GD_REQUIRE(x_rows > 0 && x_rows <= (1 << 24) && ldx * 4 < (1 << 24) && (int64_t)x_rows * ldx * 4 < (1ll << 32), GD_E_DIM,
             "gd_agg_gemm_f32: x too large for 32-bit row offsets (x_rows=%d)", x_rows);
  GD_REQUIRE(!(bias && gate_bits), GD_E_DIM, "gd_agg_gemm_f32: bias and gate are exclusive");
  GD_REQUIRE(x != y, GD_E_DIM, "gd_agg_gemm_f32: x and y must not alias");
  if (n_rows == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const int n_tiles = (n_rows + 15) / 16;
  constexpr int NW = 10;
  int grid = (n_tiles + NW - 1) / NW;
  if (grid > 256) grid = 256;                       // one block per CU: the 64 KB image + NW tiles fill its LDS
  grid = (grid + 7) / 8 * 8;
#define GD_AG_LAUNCH(DI, DO, GT)                                                                                     \
  hipLaunchKernelGGL((agg_gemm_kernel<DI, DO, NW, GT>), dim3(grid), dim3(NW * 64), 0, s, rowptr, col, val, x, ldx, rows, \
                     n_rows, w, w_out_in, bias, gate_bits, y, ldy, nnz)
#define GD_AG_CASE(DI, DO)          \
  do {                              \
    if (gate_bits) GD_AG_LAUNCH(DI, DO, true); \
    else GD_AG_LAUNCH(DI, DO, false);          \
  } while (0)
  if (d_in == 128 && d_out == 128) GD_AG_CASE(128, 128);
  else if (d_in == 128 && d_out == 64) GD_AG_CASE(128, 64);
  else if (d_in == 64 && d_out == 128) GD_AG_CASE(64, 128);
  else GD_AG_CASE(64, 64);
#undef GD_AG_CASE
#undef GD_AG_LAUNCH
  return launched("agg_gemm");
}
