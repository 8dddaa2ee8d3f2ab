// Aggregate-then-transform in ONE kernel:   y[r,:] = ( sum_k val[k] x[col[k],:] ) @ W (+ bias)   for r in rows
//
// = a GCN / GIN layer whose input is no wider than its output (framework/models/gcn.py:11-24, gin.py:26-34;
// by linearity A (x W^T) = (A x) W^T), and the layer-2 input gradient (A^T dp2)[S1] W2 with the ReLU gate.
// Run separately, the aggregation is bound by the gather path (the matrix cores idle for ~100 us) and the
// transform by the matrix cores (the memory system idles for ~90 us), and running the two KERNELS side by side
// only makes them time-share the CUs' wave slots.  Here every wave alternates between the
// two phases on its own 16-row tile, so while some waves of a CU wait for gathered rows others feed the MFMA
// pipe; the aggregated tile never leaves the CU (registers -> 8 KB of LDS -> MFMA operand registers).
// Measured (DESIGN.md sections 7 and 9): layer 1 in 187 us against 205 us apart - not max(gather, MFMA), because
// the gather path slows down with the clock once the matrix cores are busy.
//
//   phase A  row-set walk: the G = 64 / LPR lane groups (LPR = d_in / 4 lanes x float4 cover a row) take G
//            consecutive rows of the tile, one row each, and advance together - no reduction across groups, each
//            group writes its own LDS row.  Batches of U = 8 gather instructions, all unconditional (padded trips
//            re-read a cached row with weight 0) so that the s_waitcnt in front of a batch's consumer can leave
//            the next batch in flight; two batches rotate.  Few waves per CU (the 64 KB weight image limits them
//            to NW) but up to 16 loads in flight per wave, which the gather probe shows is enough
//            (tools/experiments/gather_probe.hip: 8 waves/CU x 8 loads reach the rate of 32 waves/CU).
//   phase B  D^T[feat][sample] = W[feat][k] . X^T[k][sample] with v_mfma_f32_16x16x4_f32: lane (n = l % 16,
//            q = l / 16) holds sample n's features k in [KQ q, KQ q + KQ) (8 or 4 ds_read_b128 from the tile,
//            conflict-free with the +4 pitch) and, per output tile, the weight image pre-permuted so that its
//            operand is one ds_read_b128 per 4 MFMAs with consecutive lanes on consecutive 16 bytes.  Both
//            operands use the same k permutation (k = KQ q + 4 j + c at step (j, c)), which a dot product does
//            not see.  The lane ends with 4 consecutive output features of its sample: one 16-byte store.
//
// A row is walked by ONE wave, so heavy rows would serialise.  Two ways around that: the work-item form takes
// the balanced SpMM's items (graph.py SplitPlan: hub rows cut into <= 64-edge pieces) - a piece's product goes
// to an extra output row and a fix-up adds the pieces of a row up (the transform is linear) - and the row-list
// form (used with the gate) expects hub rows pre-aggregated into extra rows of x (graph.py CappedCSR).
#include "common.h"

namespace gd {

using v4f = __attribute__((ext_vector_type(4))) float;

template <int DIN, int DOUT, int NW, bool GATE>
__global__ __launch_bounds__(NW * 64) void agg_gemm_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rows, int32_t n_rows,
    const float* __restrict__ w, int32_t w_out_in, const float* __restrict__ bias,
    const uint32_t* __restrict__ gate_bits, float* __restrict__ y, int64_t ldy, int32_t nnz, const int4* __restrict__ items, int32_t piece_base) {
  constexpr int LPR = DIN / 4;             // lanes per gathered row
  constexpr int G = kWave / LPR;           // rows per gather instruction
  constexpr int U = 8;                     // gather instructions per batch
  constexpr int PITCH = DIN + 4;           // tile row pitch in floats
  constexpr int KQ = DIN / 4;              // features per lane group q in phase B
  constexpr int J = KQ / 4;                // float4 operand fragments per lane
  constexpr int TOUT = DOUT / 16;          // 16-wide output tiles
  constexpr int kXcd = 8;
  __shared__ __attribute__((aligned(16))) float wimg[DIN * DOUT];
  __shared__ __attribute__((aligned(16))) float tiles[NW][16 * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- weight image: wimg[((t J + j) 64 + l) 4 + c] = W[feat = 16 t + l % 16][k = KQ (l / 16) + 4 j + c]
  if (w_out_in) {
    // [feat][k] source: 4 consecutive k are one image float4 - coalesced 16-byte reads, one round trip
    constexpr int K4 = DIN / 4;
    for (int e4 = tid; e4 < DOUT * K4; e4 += NW * 64) {
      const int feat = e4 / K4, k = 4 * (e4 % K4);
      const int t = feat >> 4, l = (k / KQ) * 16 + (feat & 15), j = (k % KQ) >> 2;
      reinterpret_cast<float4*>(wimg)[(t * J + j) * 64 + l] = reinterpret_cast<const float4*>(w)[e4];
    }
  } else {
    for (int e = tid; e < DIN * DOUT; e += NW * 64) {
      const int c = e & 3, l = (e >> 2) & 63, tj = e >> 8;
      const int j = tj % J, t = tj / J;
      const int feat = 16 * t + (l & 15), k = KQ * (l >> 4) + 4 * j + c;
      wimg[e] = w[k * DOUT + feat];
    }
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
  // Tile assignment: the XCD's contiguous range of tiles is cut into one contiguous chunk per block (consecutive
  // tiles share gathered rows in that XCD's L2) and the block's waves draw tiles from the chunk through a counter
  // in LDS, so a wave that drew a heavy tile (hub pieces, long rows) simply draws fewer.  (One counter per XCD in
  // global memory measured 35 % slower - 320 waves on one atomic; visiting heaviest-first 6 % slower - the order
  // is the locality.)
  __shared__ int next_tile;
  if (tid == 0) next_tile = 0;
  __syncthreads();
  const int blocks_per_xcd = gridDim.x / kXcd, bx = blockIdx.x / kXcd;
  const int n_x = t_last - t_first;
  const int chunk = (n_x + blocks_per_xcd - 1) / blocks_per_xcd;
  const int c_first = t_first + bx * chunk, c_last = min(t_last, c_first + chunk);
  auto draw = [&]() {
    int t = 0;
    if (lane == 0) t = atomicAdd(&next_tile, 1);
    return c_first + __builtin_amdgcn_readfirstlane(t);
  };
  for (int tl = draw(); tl < c_last; tl = draw()) {
    // ================= phase A: aggregate the tile's 16 rows into LDS
    // lane i < 16 holds row i's id and edge range (rows past the end are empty)
    const int slot = 16 * tl + (lane & 15);
    const bool live = slot < n_rows;
    int rid = 0, rs = 0, re = 0;
    bool biased = true;
    if (items) {
      // work items of the balanced SpMM (graph.py SplitPlan): a whole row, or one <= 64-edge piece of a hub row
      // whose product goes to the extra output row piece_base + slot (no bias) for gd's fix-up to add up
      if (live) {
        const int4 it = items[slot];
        rs = it.y;
        re = it.z;
        biased = it.w < 0;
        rid = biased ? it.x : piece_base + it.w;
      }
    } else if (live) {
      rid = rows ? rows[slot] : slot;
      rs = rowptr[rid];
      re = rowptr[rid + 1];
    }

    // Row-set walk: the G lane groups take G consecutive rows of the tile (one row each - no reduction across
    // groups, each group writes its own LDS row) and advance together, trip T of the set gathers edge T of each
    // row (groups whose row is shorter re-read a cached row with weight 0).  EVERY batch issues exactly U gathers
    // and one (col, val) prefetch, unconditionally: with a fixed number of loads per batch the s_waitcnt in front
    // of a batch's consumer can leave the NEXT batch's loads in flight - with conditional loads the compiler has
    // to wait for everything and the pipeline collapses to one batch.  Two batches rotate: up to 2 U loads in
    // flight per wave, which is what lets 10 waves per CU reach the gather rate of 32.
    // (c, w) hold LPR consecutive edges of each group's row (a "segment"); `seg` = which one.
    constexpr int NSET = 16 / G;
    int set = 0, T = 0;                                   // scalar: current row set, trips done in it
    auto set_rows = [&](int st, int& s0, int& n0) {       // per lane group: edge range of its row in set st
      const int r = min(st, NSET - 1) * G + g;
      s0 = __shfl(rs, r);
      n0 = st < NSET ? __shfl(re, r) - s0 : 0;
    };
    auto set_max = [&](int n0) {                          // longest row of the set (scalar)
      int m = n0;
#pragma unroll
      for (int off = LPR; off < kWave; off <<= 1) m = max(m, __shfl_xor(m, off));
      return __builtin_amdgcn_readfirstlane(m);
    };
    auto load_seg = [&](int s0, int sg, int& c0, float& w0) {
      const int k = min(s0 + sg * LPR + li, nnz - 1);
      c0 = col[k];
      w0 = val[k];
    };
    int s_g, n_g, s_n, n_n;                               // this set's / the next set's row per group
    set_rows(0, s_g, n_g);
    set_rows(1, s_n, n_n);
    int max_cur = set_max(n_g), max_nxt = set_max(n_n);
    int seg = 0;
    int c_cur, c_nxt;
    float w_cur, w_nxt;
    load_seg(s_g, 0, c_cur, w_cur);
    // prefetch target: the next segment of this set if its longest row needs one, else the next set's first
    bool nxt_same = max_cur > LPR;
    if (nxt_same) load_seg(s_g, 1, c_nxt, w_nxt); else load_seg(s_n, 0, c_nxt, w_nxt);
    struct Batch { float4 xv[U]; float wj[U]; int set; bool last; };
    auto issue = [&](Batch& b) {
      b.set = set;
      const int base = T - seg * LPR;                      // trip offset inside the segment
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = base + u;                            // edge slot inside the segment (< LPR)
        const int src4 = 4 * (g * LPR + min(j, LPR - 1));
        const int cs = __builtin_amdgcn_ds_bpermute(src4, c_cur);
        const float wv = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(w_cur)));
        b.wj[u] = (T + u) < n_g ? wv : 0.f;
        b.xv[u] = *reinterpret_cast<const float4*>(xb + (__umul24((uint32_t)cs, pitch_b) + lo));
      }
      T += U;
      const bool set_done = T >= max_cur;
      b.last = set_done;
      const bool seg_done = set_done || (T - seg * LPR) >= LPR;
      if (seg_done) {
        c_cur = c_nxt;
        w_cur = w_nxt;
        if (set_done) {
          set += 1;
          T = 0;
          seg = 0;
          s_g = s_n;
          n_g = n_n;
          max_cur = max_nxt;
          set_rows(set + 1, s_n, n_n);
          max_nxt = set_max(n_n);
        } else {
          seg += 1;
        }
        nxt_same = max_cur > (seg + 1) * LPR;
      }
      // always refresh the prefetch (fixed load count per batch)
      if (nxt_same) load_seg(s_g, seg + 1, c_nxt, w_nxt); else load_seg(s_n, 0, c_nxt, w_nxt);
    };
    float4 acc = f4_zero();
    auto consume = [&](const Batch& b) {
#pragma unroll
      for (int u = 0; u < U; ++u) acc = f4_fma(b.wj[u], b.xv[u], acc);
      if (b.last) {
        *reinterpret_cast<float4*>(tile + (b.set * G + g) * PITCH + 4 * li) = acc;
        acc = f4_zero();
      }
    };
    // every path below is issue -> consume in straight-line code so that the waits count exactly
    Batch b0, b1;
    issue(b0);
    while (true) {
      if (set >= NSET) { consume(b0); break; }
      issue(b1);
      consume(b0);
      if (set >= NSET) { consume(b1); break; }
      issue(b0);
      consume(b1);
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
        dacc[t] = biased ? v4f{bv.x, bv.y, bv.z, bv.w} : v4f{0.f, 0.f, 0.f, 0.f};
      } else {
        dacc[t] = v4f{0.f, 0.f, 0.f, 0.f};
      }
    }
    const float4* wi = reinterpret_cast<const float4*>(wimg) + lane;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      // the weight fragments of ONE k group at a time (the compiler barrier keeps the scheduler from hoisting all
      // J x TOUT ds_reads - 256 registers - above the first MFMA; requesting the next group before this group's
      // MFMAs measured the same and costs 32 registers)
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

// y[row,:] = sum of the row's piece products (extra rows piece_base + first_slot ...) + bias, in slot order
__global__ __launch_bounds__(256) void agg_gemm_fixup_kernel(const int4* __restrict__ split, int32_t n_split,
                                                             float* __restrict__ y, int64_t ldy, int32_t piece_base,
                                                             const float* __restrict__ bias, int32_t d4) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_split) return;
  const int4 sp = split[i];
  for (int vec = lane; vec < d4; vec += kWave) {
    float4 o = f4_zero();
    for (int sl = 0; sl < sp.z; ++sl)
      o = f4_add(o, reinterpret_cast<const float4*>(y + (int64_t)(piece_base + sp.y + sl) * ldy)[vec]);
    if (bias) o = f4_add(o, reinterpret_cast<const float4*>(bias)[vec]);
    reinterpret_cast<float4*>(y + (int64_t)sp.x * ldy)[vec] = o;
  }
}

}  // namespace gd

extern "C" int gd_agg_gemm_f32(const int32_t* rowptr, const int32_t* col, const float* val, const float* x, int64_t ldx,
                               int32_t x_rows, const int32_t* rows, int32_t n_rows, const float* w, int32_t d_in,
                               int32_t d_out, int32_t w_out_in, const float* bias, const uint32_t* gate_bits, float* y,
                               int64_t ldy, int32_t nnz, const int32_t* items, const int32_t* split, int32_t n_split, int32_t piece_base,
                               void* stream) {
  using namespace gd;
  GD_REQUIRE((rowptr || items) && col && val && x && w && y, GD_E_NULL, "gd_agg_gemm_f32: null pointer");
  GD_REQUIRE(!items || (!rows && !gate_bits && aligned16(items) && (n_split == 0 || (split && aligned16(split)))), GD_E_DIM,
             "gd_agg_gemm_f32: the work-item form takes no row list / gate and needs 16-byte aligned items");
  GD_REQUIRE(nnz > 0, GD_E_DIM, "gd_agg_gemm_f32: nnz must be the length of col (> 0)");
  GD_REQUIRE((d_in == 64 || d_in == 128) && (d_out == 64 || d_out == 128), GD_E_DIM,
             "gd_agg_gemm_f32: d_in=%d d_out=%d must each be 64 or 128", d_in, d_out);
  GD_REQUIRE(n_rows >= 0 && ldx >= d_in && ldy >= d_out && ldx % 4 == 0 && ldy % 4 == 0, GD_E_DIM,
             "gd_agg_gemm_f32: bad leading dimensions");
  GD_REQUIRE(aligned16(x) && aligned16(y) && aligned16(w) && (!bias || aligned16(bias)), GD_E_ALIGN, "gd_agg_gemm_f32: unaligned pointer");
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
                     n_rows, w, w_out_in, bias, gate_bits, y, ldy, nnz, reinterpret_cast<const int4*>(items), piece_base)
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
  int rc = launched("agg_gemm");
  if (rc || !items || n_split == 0) return rc;
  hipLaunchKernelGGL(agg_gemm_fixup_kernel, dim3((n_split + 3) / 4), dim3(256), 0, s, reinterpret_cast<const int4*>(split),
                     n_split, y, ldy, piece_base, bias, d_out / 4);
  return launched("agg_gemm_fixup");
}
