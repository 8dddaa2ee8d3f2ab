// R-GCN typed message passing, wave-private form (PyG RGCNConv aggr='mean' with constant BLOCK-DIAGONAL relation weights,
// framework/models/rgcn.py:16-38; same contract as gd_rgcn_conv_f32 / gd_rgcn_tile_conv_f32):
//
//     y[i,:] += sum_r ( sum_{e in run(i,r)} w_e x[col_e,:] ) @ W_r
//
// The tile kernel (rgcn_tile.hip) walks (64-node tile, relation) steps with a block of 8 waves and two block barriers
// per step: every step is a dependent chain descriptor -> edge list -> gathered rows -> LDS tile -> barrier -> matrix
// instructions -> barrier, and the waves wait 67 % of their cycles (profiles/r03_rgcn_tile_issue_pmc.txt).  With the
// reference's four diagonal blocks, output block t of a node depends on input features [KL t, KL t + KL) only - so the
// FOUR BLOCKS ARE FOUR INDEPENDENT PROBLEMS and a single wave can own one of them for a whole tile:
//
//   job    = (tile of TILE consecutive nodes, diagonal block t); one wave, no barrier, no atomics on global memory; the
//            job's outputs [TILE x OW] live in a wave-private LDS accumulator across all relations of the tile;
//   unit   = 16 pieces of ONE relation of the tile, piece = up to 4 in-edges of one (node, relation) run, in a fixed
//            shape: 16 slots x 4 (source, weight) pairs, unused pairs point one row past x (the buffer descriptor
//            returns zeros for them without touching memory) - so that every unit issues the SAME number of loads and
//            every wait in the loop is an exact count;
//   gather   LPR = KL / 4 lanes take the 16-byte pieces of one source row's [KL t, KL t + KL) slice (128 B = one cache
//            line at KL = 32), 64 / LPR slots per load instruction, 4 loads per slot; the weighted sum of a slot is the
//            compact row q of a [16 x KL] tile in LDS;
//   product  D[out][slot] = W_r^T[out][k] A^T[k][slot] on v_mfma_f32_16x16x4_f32 with the relation's block from
//            gd_rgcn_pack_weight_f32 (the same packed image the tile kernel reads).  A run longer than 4 edges is several
//            CONSECUTIVE slots with the same node row; the slots of a unit are the 16 lanes of a DPP row of D, so a
//            segmented scan over those lanes (v_mov_dpp row_shr:1 / 2 / 4 / 8 + FMA per register; the segment flags come with
//            the plan, which also says per unit WHICH of the four steps any of its slots needs - most need one or two)
//            leaves a row's total in its LAST slot, and only that lane adds it to the node row of the accumulator: a plain
//            LDS read-modify-write without two lanes on one address.  (LDS float atomics would not need the scan - and were
//            measured at ~180 cycles per instruction: the LDS pipe 86 % busy, 3.9 ms per launch.)
//   pipeline the rows of the units u + 1 .. u + DEPTH, the slot words of unit u + 1 and the edge pairs of unit u + DEPTH + 2
//            (ONE 8-byte load per lane, staged through LDS a pass later) are in flight while unit u is summed and multiplied
//            (DEPTH + 1 register sets, the loop is unrolled DEPTH + 1 times; past the tile's end the passes run on the plan's
//            empty unit); the weight fragments of unit u + 1 replace unit u's right after its products - only when the
//            relation changes (a uniform branch around four loads: the waits stay exact counts).
//
// Measured (layer-1 launch of the biokg request, NOTES round 4): LDS atomics 3.9 ms -> scan 0.94 ms -> edge pairs through LDS,
// conditional weight reload, per-unit scan steps 0.72 ms (tile kernel: 1.22 ms).  The texture addresser was the limiter on the way
// (17 loads x 16 cycles per unit); now the gathered volume (every source row once per edge: 4.0-4.5 GB, all of it fabric traffic,
// moved at 5.5-6 TB/s) bounds the launch, with the on-CU work ~10 % behind (650 us with cache-resident sources).
#include <stdlib.h>

#include "common.h"

namespace gd {

using f32x4w = __attribute__((ext_vector_type(4))) float;
using u32x4w = __attribute__((ext_vector_type(4))) unsigned int;

template <int KL, int OW, int TILE, int DEPTH, int BPW, bool RELU>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DEPTH == 1 ? 3 : 2))) void rgcn_wave_kernel(
    const int32_t* __restrict__ job_tile, int32_t n_tiles, const int32_t* __restrict__ tile_unit_ptr,
    const int32_t* __restrict__ unit_rel, const int4* __restrict__ unit_edges, const int32_t* __restrict__ unit_row,
    const float* __restrict__ x, int64_t ldx, const float4* __restrict__ wpk4, float* __restrict__ y, int64_t ldy,
    int32_t n_nodes, int64_t n_x_bytes, int32_t empty_unit, int32_t xcd_blocks) {
  // BPW diagonal blocks per wave (a job is (tile, BPW consecutive blocks); 2 exists for 16-wide blocks as an opt-in, see the launch)
  constexpr int KW = BPW * KL;                             // gathered floats per source row
  constexpr int LPR = KW / 4, GROUPS = 64 / LPR, ROUNDS = 16 / GROUPS, AP = KW + 4, NOH = OW / 16, NMM = KL / 16, CP = BPW * OW + 4;
  constexpr int JPT = 4 / BPW;                             // jobs per tile
  __shared__ __attribute__((aligned(16))) float a_tile[16 * AP];
  __shared__ __attribute__((aligned(16))) float acc[TILE * CP];      // [node row][output], pitch OW + 4
  __shared__ __attribute__((aligned(16))) int e_lds[128];             // one unit's 64 (source, weight) pairs
  const int lane = threadIdx.x, b = blockIdx.x;
  // b = 8 JPT q + 8 t + xcd: the jobs of a tile run on the same XCD (workgroups go round the XCDs), next to each other
  int ot = BPW * ((b >> 3) % JPT), ti = ((b / (8 * JPT)) << 3) + (b & 7);     // ot = the job's first diagonal block
  if (BPW == 1 && xcd_blocks) {
    // diagonal block t on the XCD pair (2 t, 2 t + 1): an L2 then serves the 128-byte column slice t of
    // the source rows - a quarter of the table - against reading every tile's unit plan on four XCDs instead of one.
    // Measured on the biokg request (profiles/r05_rgcn_xcd_blocks.txt): -22 % / -19 % fabric traffic for the two launches
    // with 128-float sources, +57 % for the 64-float one (the launch code picks per width).
    const int xcd = b & 7;
    ot = xcd >> 1;
    ti = ((b >> 3) << 1) + (xcd & 1);
  }
  if (ti >= n_tiles) return;
  const int tile = __builtin_amdgcn_readfirstlane(job_tile ? job_tile[ti] : ti);
  const int u0 = __builtin_amdgcn_readfirstlane(tile_unit_ptr[tile]), u1 = __builtin_amdgcn_readfirstlane(tile_unit_ptr[tile + 1]);
  if (u0 >= u1) return;
  const int g = lane / LPR, gl = lane % LPR, j = lane & 15, kq = lane >> 4;
  for (int i = lane; i < TILE * CP / 4; i += 64) reinterpret_cast<float4*>(acc)[i] = f4_zero();

  const uint32_t row_bytes = (uint32_t)(ldx * 4), feat_off = (uint32_t)(4 * (ot * KL + 4 * gl));
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (uint32_t)n_x_bytes, 0x00020000);
  // Edge pairs of a unit: ONE 8-byte load per lane (the unit's 64 pairs in slot order: 512 contiguous bytes, 8 cycles of the
  // texture addresser) staged through a wave-private LDS image, from which lane group g reads the four pairs of its slot
  // of each round (two broadcast 16-byte reads).  Fetching them as 16-byte global loads cost 4 x 16 addresser cycles per
  // unit - the addresser was 73 % busy and bounded the kernel (profiles/r04_rgcn_wave_pmc.txt).
  auto fetch_edges = [&](int u) -> int2 { return reinterpret_cast<const int2*>(unit_edges)[(int64_t)u * 64 + lane]; };
  auto stage_edges = [&](int2 ge, int4 (&e)[ROUNDS][2]) {
    reinterpret_cast<int2*>(e_lds)[lane] = ge;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int4* src = reinterpret_cast<const int4*>(e_lds) + (GROUPS * r + g) * 2;
      e[r][0] = src[0];
      e[r][1] = src[1];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto issue_rows = [&](const int4 (&e)[ROUNDS][2], float4 (&rows)[ROUNDS][4], float (&wt)[ROUNDS][4]) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int cc[4] = {e[r][0].x, e[r][0].z, e[r][1].x, e[r][1].z};
      wt[r][0] = __int_as_float(e[r][0].y);
      wt[r][1] = __int_as_float(e[r][0].w);
      wt[r][2] = __int_as_float(e[r][1].y);
      wt[r][3] = __int_as_float(e[r][1].w);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, __umul24((uint32_t)cc[k], row_bytes) + feat_off, 0, 0);
        rows[r][k] = __builtin_bit_cast(float4, v);
      }
    }
  };
  // rel = the unit's relation, read a unit ahead (a scalar load: its latency must not sit in front of the weight loads)
  auto load_wf = [&](int rel, float4 (&wf)[BPW][NOH][NMM]) {
#pragma unroll
    for (int bw = 0; bw < BPW; ++bw) {
      const float4* wp = wpk4 + ((int64_t)(rel * 4 + ot + bw) * (NOH * NMM)) * 64 + lane;
#pragma unroll
      for (int oh = 0; oh < NOH; ++oh)
#pragma unroll
        for (int mm = 0; mm < NMM; ++mm) wf[bw][oh][mm] = wp[(oh * NMM + mm) * 64];
    }
  };
  // rel_cur / rel_next: unit_rel words (relation | scan steps << 16) of this and the next unit
  auto compute = [&](const float4 (&rows)[ROUNDS][4], const float (&wt)[ROUNDS][4], float4 (&wf)[BPW][NOH][NMM], int nrow, int rel_cur, int rel_next) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      // RELU: the conv reads relu(x) (RGCN's layer 2 reads relu(z1), rgcn.py:36-37) - formed here, where the rows land, instead
      // of in a pass of its own over [n, d] (a torch clamp kernel, 15 us of the biokg step; the launch is fabric-bound)
      auto in = [&](const float4& v) { return RELU ? make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)) : v; };
      const float4 v0 = in(rows[r][0]);
      float4 s = make_float4(wt[r][0] * v0.x, wt[r][0] * v0.y, wt[r][0] * v0.z, wt[r][0] * v0.w);
#pragma unroll
      for (int k = 1; k < 4; ++k) s = f4_fma(wt[r][k], in(rows[r][k]), s);
      *reinterpret_cast<float4*>(a_tile + (GROUPS * r + g) * AP + 4 * gl) = s;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lane (j, kq) feeds k = 16 mm + 4 kq + c of block bw of slot j and ends with the outputs 16 oh + 4 kq + c of that block
    float4 bv[BPW][NMM];
#pragma unroll
    for (int bw = 0; bw < BPW; ++bw)
#pragma unroll
      for (int mm = 0; mm < NMM; ++mm) bv[bw][mm] = *reinterpret_cast<const float4*>(a_tile + j * AP + bw * KL + 16 * mm + 4 * kq);
    f32x4w d[BPW][NOH];
#pragma unroll
    for (int bw = 0; bw < BPW; ++bw)
#pragma unroll
      for (int oh = 0; oh < NOH; ++oh) d[bw][oh] = f32x4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mm = 0; mm < NMM; ++mm) {
#pragma unroll
      for (int bw = 0; bw < BPW; ++bw)
#pragma unroll
        for (int oh = 0; oh < NOH; ++oh) {
          d[bw][oh] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[bw][oh][mm].x, bv[bw][mm].x, d[bw][oh], 0, 0, 0);
          d[bw][oh] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[bw][oh][mm].y, bv[bw][mm].y, d[bw][oh], 0, 0, 0);
          d[bw][oh] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[bw][oh][mm].z, bv[bw][mm].z, d[bw][oh], 0, 0, 0);
          d[bw][oh] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[bw][oh][mm].w, bv[bw][mm].w, d[bw][oh], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the next unit's weight fragments replace this unit's - unless it is the same relation (2.2 units per (tile, relation)
    // on the biokg request: more than half of the reloads, 4 x 16 addresser cycles each, fall away)
    if ((rel_next & 0xffff) != (rel_cur & 0xffff)) load_wf(rel_next & 0xffff, wf);
    __builtin_amdgcn_sched_barrier(0);
    // nrow = node row | segment flags << 8 | last-of-its-row << 12.  Flag bit b: the slot 2^b to the left holds the same
    // node row (same-row slots are consecutive) - Hillis-Steele steps 1, 2, 4, 8 inside the 16-lane DPP row; the unit's
    // word says which steps any of its slots needs (bit 16 + b): most units need the first one or two only
#define GD_RW_SCAN_STEP(B, CTRL)                                                                                   \
    if (rel_cur & (1 << (16 + B))) {                                                                                 \
      const float f = (float)((nrow >> (8 + B)) & 1);                                                                \
      _Pragma("unroll") for (int bw = 0; bw < BPW; ++bw)                                                             \
        _Pragma("unroll") for (int oh = 0; oh < NOH; ++oh)                                                           \
          _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                              \
            d[bw][oh][c] = fmaf(f, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d[bw][oh][c]), CTRL, 0xF, 0xF, true)), d[bw][oh][c]); \
    }
    GD_RW_SCAN_STEP(0, 0x111)
    GD_RW_SCAN_STEP(1, 0x112)
    GD_RW_SCAN_STEP(2, 0x114)
    GD_RW_SCAN_STEP(3, 0x118)
#undef GD_RW_SCAN_STEP
    if (nrow & (1 << 12)) {
      float* dst = acc + (nrow & 255) * CP + 4 * kq;
#pragma unroll
      for (int bw = 0; bw < BPW; ++bw)
#pragma unroll
        for (int oh = 0; oh < NOH; ++oh) {
          float4 v = *reinterpret_cast<float4*>(dst + bw * OW + 16 * oh);
          v.x += d[bw][oh][0]; v.y += d[bw][oh][1]; v.z += d[bw][oh][2]; v.w += d[bw][oh][3];
          *reinterpret_cast<float4*>(dst + bw * OW + 16 * oh) = v;
        }
    }
  };

  // DEPTH units of rows in flight per wave (DEPTH + 1 register sets, the loop unrolled DEPTH + 1 times; DEPTH odd so that
  // the two edge sets alternate consistently).  Units past the tile's end are the plan's EMPTY unit (all pairs one row
  // past x, slot words 0): they cost a pass of the loop body and add nothing.
  constexpr int NS = DEPTH + 1;
  static_assert(DEPTH % 2 == 1, "DEPTH must be odd");
  int4 e[2][ROUNDS][2];
  float4 rows[NS][ROUNDS][4], wf[BPW][NOH][NMM];
  float wt[NS][ROUNDS][4];
  int nrow[2];
  auto uid = [&](int v) { return v < u1 ? v : empty_unit; };
  // Order inside a pass (unit u): the slot words of unit u + 1 FIRST (the packed FMAs of the sums read register PAIRS; when
  // the allocator pairs a weight with the target of a load in flight, the wait in front of that FMA covers everything issued
  // up to that load - first in the pass, that is only what the sums need anyway), then the edge pairs of unit u + DEPTH + 2
  // (global -> register), the pairs of unit u + DEPTH + 1 (fetched a pass ago) through LDS into their register set, then
  // the rows of unit u + DEPTH; the next unit's weight fragments replace this unit's right after its products.
  nrow[0] = unit_row[u0 * 16 + j];
  int rel_c = __builtin_amdgcn_readfirstlane(unit_rel[u0]), rel_n = __builtin_amdgcn_readfirstlane(unit_rel[uid(u0 + 1)]);
  load_wf(rel_c & 0xffff, wf);
  {
    int2 g0 = fetch_edges(u0);
    stage_edges(g0, e[0]);
  }
#pragma unroll
  for (int k = 0; k < DEPTH; ++k) {
    int2 gk = fetch_edges(uid(u0 + k + 1));
    issue_rows(e[k & 1], rows[k], wt[k]);
    stage_edges(gk, e[(k + 1) & 1]);
  }
  int2 ge = fetch_edges(uid(u0 + DEPTH + 1));                // e[DEPTH & 1] holds unit u0 + DEPTH, ge unit u0 + DEPTH + 1
  for (int u = u0; u < u1; u += NS) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      nrow[(i + 1) & 1] = unit_row[uid(u + i + 1) * 16 + j];
      const int rel_nn = __builtin_amdgcn_readfirstlane(unit_rel[uid(u + i + 2)]);
      const int2 ge_new = fetch_edges(uid(u + i + DEPTH + 2));
      issue_rows(e[(i + 1) & 1], rows[(i + DEPTH) % NS], wt[(i + DEPTH) % NS]);
      stage_edges(ge, e[i & 1]);                             // unit u + i + DEPTH + 1, for the next pass
      ge = ge_new;
      __builtin_amdgcn_sched_barrier(0);
      compute(rows[i], wt[i], wf, nrow[i & 1], rel_c, rel_n);
      __builtin_amdgcn_sched_barrier(0);
      rel_c = rel_n;
      rel_n = rel_nn;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // the tile's rows of this job's outputs: y += accumulators (BPW OW / 4 lanes per row)
  constexpr int C4 = BPW * OW / 4;
  for (int i = lane; i < TILE * C4; i += 64) {
    const int r = i / C4, c4 = i % C4, node = tile * TILE + r;
    if (node < n_nodes) {
      const float4 a = *reinterpret_cast<const float4*>(acc + r * CP + 4 * c4);
      float4* dst = reinterpret_cast<float4*>(y + (int64_t)node * ldy + OW * ot) + c4;
      *dst = f4_add(*dst, a);
    }
  }
}

static bool wave_geometry(int32_t d_in, int32_t d_out, int32_t n_blocks) {
  return n_blocks == 4 && (d_in == 64 || d_in == 128) && (d_out == 64 || d_out == 128);
}

}  // namespace gd

extern "C" int32_t gd_rgcn_wave_covers(int32_t d_in, int32_t d_out, int32_t n_blocks) {
  return gd::wave_geometry(d_in, d_out, n_blocks) ? 1 : 0;
}

extern "C" int gd_rgcn_wave_conv_f32(const int32_t* job_tile, int32_t n_tiles, int32_t tile, const int32_t* tile_unit_ptr,
                                     int32_t n_units, const int32_t* unit_rel, const int32_t* unit_edges, const int32_t* unit_row,
                                     const float* x, int64_t ldx, int32_t d_in, const float* packed_w, int32_t n_blocks, float* y,
                                     int64_t ldy, int32_t d_out, int32_t n_nodes, int32_t relu_in, void* stream) {
  using namespace gd;
  GD_REQUIRE(tile_unit_ptr && unit_rel && unit_edges && unit_row && x && packed_w && y, GD_E_NULL, "gd_rgcn_wave_conv_f32: null pointer");
  GD_REQUIRE(wave_geometry(d_in, d_out, n_blocks), GD_E_DIM,
             "gd_rgcn_wave_conv_f32: needs 4 diagonal blocks and widths in {64, 128} (d_in=%d d_out=%d blocks=%d); use gd_rgcn_tile_conv_f32",
             d_in, d_out, n_blocks);
  GD_REQUIRE(tile == 64 && n_tiles == (n_nodes + tile - 1) / tile && n_units >= 0 && ldx >= d_in && ldy >= d_out && ldx % 4 == 0 &&
                 ldy % 4 == 0, GD_E_DIM,
             "gd_rgcn_wave_conv_f32: tile must be 64, n_tiles = ceil(n_nodes / tile), row pitches multiples of 4");
  GD_REQUIRE(aligned16(x) && aligned16(y) && aligned16(packed_w) && aligned16(unit_edges) && x != y, GD_E_ALIGN,
             "gd_rgcn_wave_conv_f32: unaligned or aliasing pointer");
  // 24 x 24-bit row offsets; the pad source n_nodes (one row past x) must stay below 4 GB as well
  GD_REQUIRE(n_nodes < (1 << 24) && ldx * 4 < (1 << 24) && ((int64_t)n_nodes + 1) * ldx * 4 < ((int64_t)1 << 32), GD_E_DIM,
             "gd_rgcn_wave_conv_f32: x beyond 4 GB / 2^24 rows; use gd_rgcn_conv_f32");
  if (n_tiles == 0 || n_units == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const unsigned tile_groups = (unsigned)((n_tiles + 7) / 8);
  const int64_t n_x_bytes = ((int64_t)(n_nodes - 1) * ldx + d_in) * 4;
  // units of gathered rows in flight per wave: 1 (three waves per SIMD) for 128-float sources, 3 for 64-float sources (their
  // units carry half the bytes; measured on the biokg request: 724 vs 783 us and 468 vs 475 us); GD_RGCN_WAVE_DEPTH = 1 | 3
  // overrides (read per call: an A/B switch, not a tuning cache)
  const char* env_depth = getenv("GD_RGCN_WAVE_DEPTH");
  const int depth = env_depth ? (atoi(env_depth) == 1 ? 1 : 3) : (d_in == 128 ? 1 : 3);
  // 16-wide blocks (64-float sources): GD_RGCN_WAVE_BPW=2 gives a wave TWO blocks, so that it gathers whole 128-byte lines
  // instead of halves - measured slower on the biokg request (485 vs 423 us: 20 KB of LDS per wave leave 7 waves per CU;
  // the sibling's half line hits the L2 anyway: FETCH_SIZE is at the gathered volume), so one block per wave stays the default
  const char* env_bpw = getenv("GD_RGCN_WAVE_BPW");
  const int bpw16 = env_bpw && atoi(env_bpw) == 2 ? 2 : 1;
  // diagonal block t on the XCD pair (2 t, 2 t + 1): default for 128-float sources (a wave gathers whole 128-byte lines of its
  // column slice; FETCH_SIZE of the biokg layer-1 launch 3.99 -> 3.10 GB), not for 64-float sources (64-byte half lines whose
  // other half the sibling wave on the same XCD used to share: 1.99 -> 3.14 GB).  GD_RGCN_WAVE_XCD_BLOCKS = 0 | 1 overrides (A/B).
  const char* env_xb = getenv("GD_RGCN_WAVE_XCD_BLOCKS");
  const int xcd_blocks = env_xb ? (atoi(env_xb) == 1 ? 1 : 0) : (d_in == 128 ? 1 : 0);
  const unsigned pair_groups = (unsigned)((n_tiles + 1) / 2);
#define GD_RW_LAUNCH1(KL, OW, DEPTH, BPW, RELU)                                                                              \
  hipLaunchKernelGGL((rgcn_wave_kernel<KL, OW, 64, DEPTH, BPW, RELU>),                                                        \
                     dim3((BPW == 1 && xcd_blocks) ? pair_groups * 8 : tile_groups * 8 * (4 / BPW)), dim3(64), 0, s, job_tile, n_tiles, \
                     tile_unit_ptr, unit_rel, reinterpret_cast<const int4*>(unit_edges), unit_row, x, ldx,                 \
                     reinterpret_cast<const float4*>(packed_w), y, ldy, n_nodes, n_x_bytes, n_units, xcd_blocks)
#define GD_RW_LAUNCH(KL, OW, DEPTH, BPW)              \
  do {                                                \
    if (relu_in) GD_RW_LAUNCH1(KL, OW, DEPTH, BPW, true); \
    else GD_RW_LAUNCH1(KL, OW, DEPTH, BPW, false);    \
  } while (0)
#define GD_RW_CASE(KL, OW)                                  \
  do {                                                      \
    if (KL == 16 && bpw16 == 2) {                           \
      if (depth == 1) GD_RW_LAUNCH(KL, OW, 1, (KL == 16 ? 2 : 1)); \
      else GD_RW_LAUNCH(KL, OW, 3, (KL == 16 ? 2 : 1));     \
    } else {                                                \
      if (depth == 1) GD_RW_LAUNCH(KL, OW, 1, 1);           \
      else GD_RW_LAUNCH(KL, OW, 3, 1);                      \
    }                                                       \
  } while (0)
  switch (d_in * 1000 + d_out) {
    case 128128: GD_RW_CASE(32, 32); break;
    case 128064: GD_RW_CASE(32, 16); break;
    case 64128: GD_RW_CASE(16, 32); break;
    case 64064: GD_RW_CASE(16, 16); break;
    default: return fail(GD_E_DIM, "gd_rgcn_wave_conv_f32: no kernel for d_in=%d d_out=%d", d_in, d_out);
  }
#undef GD_RW_CASE
#undef GD_RW_LAUNCH
#undef GD_RW_LAUNCH1
  return launched("rgcn_wave_conv");
}
