// CSR SpMM over RUN items: fixed-shape units of gathers with a segmented scan instead of one accumulator per row.
//
// Why (round 5; profiles/r05_rowsize_probe.txt, profiles/NOTES.md): the item kernel of spmm.hip spends its time in DEPENDENT ROUND
// TRIPS - a wave issues one batch of U x G gathers, waits ~2.5 us, sums, moves on - and the batches are ragged: an item is one
// row (two light rows at 64 floats), so 34 % of the slots a wave holds in flight carry nothing (7.8 in-edges per row on the
// bench graph; 173,629 batches for 1.84 M edges at 16 slots).  Time = batches / resident waves x round trip explains both
// widths to 3 % (d = 64: 60.6 us; d = 128: 105 us), while the part moves random 256-byte rows at 7.1-7.6 TB/s when every slot
// is used (probe) against the 4.3-4.7 TB/s of that kernel.  More rows per item cost one accumulator each there (4 rows: 88
// VGPRs, no gain).  Here an item is a RUN of up to 64 consecutive CSR entries covering whole rows - any number of rows - and
// the wave works through it in UNITS of S = 4 G slots (G = 64 / LPR lane groups, four trips):
//   slot s = S u + 4 g + t of the run: lane group g gathers it in trip t, so a group holds FOUR CONSECUTIVE entries;
//   inside a group the products are summed by a segmented scan along t IN the registers the gathers landed in
//     (a_t = w_t x_t + (slot t - 1 ends a row ? 0 : a_(t-1))): no accumulator per row, any number of rows per unit;
//   a row that crosses groups gets the open tails of the groups before it through a wave-private LDS table (one 16-byte
//     write, G reads per lane; which tails belong to it follows from the unit's end mask, which is wave-uniform);
//   a row that crosses units gets a register carry; a row's total sits at its LAST slot and is stored from there
//     (four buffer stores per unit, out-of-range - dropped by the buffer unit - for slots that end no row).
// Every unit issues the same four 16-byte gathers per lane (slots past the run's end get an out-of-range offset: the buffer
// unit answers zeros without a memory access), so two units are in flight per wave - unit u + 1 is requested before unit u is
// summed - with counted waits.  Rows above 64 entries are GROUPS of four member items as in spmm.hip (the four waves of a
// block sum a share each, the shares meet in LDS).  The sums are associated differently from the item kernel's (sequential
// along a group's four slots, then across groups and units in slot order): bit-reproducible, equal to fp32 rounding.
// items[i] = {row0, start, mask_lo, mask_hi}: the run covers col / val [start, start + 64 - clz(mask)); bit s of the mask =
//   slot s is the last entry of its row (row0 + number of set bits below s);
//   {row | 1 << 30, start, end, 0}: member of a group (one of four consecutive, 4-aligned items);  row0 < 0: padding;
//   {row0 | 1 << 29, count, 0, 0}: `count` (<= 64) consecutive rows without entries (bias + self term only).
#include <stdlib.h>

#include "common.h"

namespace gd {

using u32x4r = __attribute__((ext_vector_type(4))) unsigned int;

template <int LPR, bool SELF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void spmm_runs_kernel(
    const int4* __restrict__ items, const int32_t* __restrict__ xcd_bounds, const int32_t* __restrict__ col,
    const float* __restrict__ val, const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy,
    const float* __restrict__ bias, float self_coef, const float* __restrict__ xs, int32_t nnz, int32_t x_rows, uint32_t x_bytes, uint32_t y_bytes) {
  constexpr int G = kWave / LPR;          // lane groups = source rows per trip
  constexpr int S = 4 * G;                // slots per unit
  constexpr int kXcd = 8;
  constexpr uint32_t kOob = 0xfffffff0u;  // beyond any buffer of less than 4 GiB: loads return 0, stores are dropped
  // unit flags (issue side -> consume side, SGPRs)
  constexpr int F_FIRST = 1, F_LAST = 2, F_MEMBER = 4, F_EMPTY = 8, F_FINAL = 16;
  __shared__ __attribute__((aligned(16))) float4 tails[4][G][LPR];
  __shared__ __attribute__((aligned(16))) float4 grp_red[4][LPR];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane / LPR, li = lane % LPR;
  const int xcd = blockIdx.x % kXcd;
  const int stride = (gridDim.x / kXcd) * 4;                        // waves per XCD
  const int wx = (blockIdx.x / kXcd) * 4 + wave;
  const int i0 = xcd_bounds[xcd], i1 = xcd_bounds[xcd + 1];
  int it = i0 + wx;                                                 // the item whose units are being REQUESTED
  if (it >= i1) return;

  const uint32_t lo = 16u * (uint32_t)li, pitch_x = (uint32_t)ldx * 4u, pitch_y = (uint32_t)ldy * 4u;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xs), 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y, 0, y_bytes, 0x00020000);
  const float4 bv = bias ? *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bias) + lo) : f4_zero();
  float4(*tl)[LPR] = tails[wave];

  // The wave's work is ONE STREAM of units: the units of the chunks (64 entries) of its items, in order.  Unit k + 1 is
  // requested before unit k is summed, across chunk and item boundaries alike, so two units are in flight per wave all the
  // time (an item-by-item pipeline drains at every item: 2.3 units per item on the bench graph).  Everything a unit's sums
  // need besides its rows is wave-uniform and travels from the request to the sum in SGPRs (UnitP).
  // Descriptors: every lane loads one dword of the 16-byte record (lane l its word l & 3), UNCONDITIONALLY - a load under
  // `if (lane == 0)` makes the compiler give up counting and every wait of the loop becomes vmcnt(0).  For the same reason the
  // prefetch loads of a step (the NEXT chunk's (col, val), the descriptor two items ahead) are issued on EVERY step: the
  // same addresses again until the stream moves on, 12 addresser cycles per unit against 64 for its gathers.
  struct Desc { int row, start, z, w; };
  struct UnitP { uint32_t m16; int rowbase, flags, row0, extra, u; };
  const int32_t* iw = reinterpret_cast<const int32_t*>(items) + (lane & 3);
  auto uniform = [](int v) {
    Desc d;
    d.row = __builtin_amdgcn_readlane(v, 0);
    d.start = __builtin_amdgcn_readlane(v, 1);
    d.z = __builtin_amdgcn_readlane(v, 2);
    d.w = __builtin_amdgcn_readlane(v, 3);
    return d;
  };
  int dvq = iw[4 * (int64_t)it];
  const int dv1 = iw[4 * (int64_t)min(it + stride, i1 - 1)];
  Desc dc = uniform(dvq), dn = uniform(dv1);
  dvq = iw[4 * (int64_t)min(it + 2 * stride, i1 - 1)];
  // request-side state of the current item / chunk (SGPRs)
  int kind = 0, row0 = 0, base = 0, end = 0, cnt = 0, nu = 1, u = 0;
  bool first_of_item = true;
  uint64_t mask = 0;
  auto decode = [&]() {
    const bool pad = dc.row < 0 || it >= i1;
    const bool member = !pad && ((dc.row >> 30) & 1), empty = !pad && ((dc.row >> 29) & 1);
    kind = member ? F_MEMBER : (empty ? F_EMPTY : 0);
    row0 = dc.row & 0x1fffffff;
    mask = (pad || member || empty) ? 0ull : (((uint64_t)(uint32_t)dc.w << 32) | (uint32_t)dc.z);
    base = (pad || empty) ? 0 : dc.start;
    end = (pad || empty) ? 0 : (member ? dc.z : dc.start + (mask ? 64 - __builtin_clzll(mask) : 0));
    first_of_item = true;
  };
  auto open_chunk = [&]() {
    cnt = max(0, min(kWave, end - base));
    nu = max(1, (cnt + S - 1) / S);
    u = 0;
  };
  auto next_chunk_start = [&]() -> int {          // where the chunk AFTER the current one begins (for the prefetch)
    if (base + kWave < end) return base + kWave;
    return (dn.row >= 0 && ((dn.row >> 29) & 1)) ? 0 : dn.start;
  };
  decode();
  open_chunk();
  // (lanes past the chunk's end carry the row id x_rows - one row past x: the buffer unit answers such a gather with zeros
  //  without a memory access, and no slot needs a compare of its own)
  int kk = min(base + lane, nnz - 1);
  int c_cur = lane < cnt ? col[kk] : x_rows;
  float w_cur = val[kk], w_prev = 0.f;
  kk = min(next_chunk_start() + lane, nnz - 1);
  int c_nx = col[kk];
  float w_nx = val[kk];
  float4 carry = f4_zero();               // open tail of the item so far (a row that crosses units; a member's whole share)

  auto advance = [&]() {                  // the request side moves to the next chunk (wave-uniform)
    w_prev = w_cur;
    w_cur = w_nx;
    if (base + kWave < end) {
      base += kWave;
    } else {
      it += stride;
      dc = dn;
      dn = uniform(dvq);
      decode();
    }
    open_chunk();
    c_cur = lane < cnt ? c_nx : x_rows;
  };
  auto prefetch = [&]() {
    const int k2 = min(next_chunk_start() + lane, nnz - 1);
    c_nx = col[k2];
    w_nx = val[k2];
    dvq = iw[4 * (int64_t)min(it + 2 * stride, i1 - 1)];
  };
  // ---- request one unit: four gathers per lane (slot S u + 4 g + t of the chunk), its parameters for the sum
  auto issue = [&](float4(&xv)[4], UnitP& p) {
    p.m16 = (uint32_t)(mask >> (S * u)) & ((1u << S) - 1u);
    p.rowbase = row0 + __builtin_popcountll(mask & ((1ull << (S * u)) - 1ull));
    const bool last = (u + 1 == nu) && !(base + kWave < end);
    p.flags = (first_of_item ? F_FIRST : 0) | (last ? F_LAST : 0) | kind | ((last && it + stride >= i1) ? F_FINAL : 0);
    p.row0 = row0;
    p.extra = dc.start;
    p.u = u;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int s = S * u + 4 * g + t;
      const int cs = __builtin_amdgcn_ds_bpermute(4 * s, c_cur);
      xv[t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, __umul24((uint32_t)cs, pitch_x) + lo, 0, 0));
    }
    first_of_item = false;
    ++u;
  };
  // ---- sum one unit (its rows have landed in a[]); moved = the request side has left this unit's chunk meanwhile
  auto consume = [&](float4(&a)[4], const UnitP& p, bool moved) {
    if (p.flags & F_FIRST) carry = f4_zero();
    const float wsrc = moved ? w_prev : w_cur;
    float wv[4];                          // the slots' weights: through the LDS crossbar now (not held across the wait)
#pragma unroll
    for (int t = 0; t < 4; ++t) wv[t] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (S * p.u + 4 * g + t), __float_as_int(wsrc)));
    const uint32_t m16 = p.m16;                                 // the unit's end bits (wave-uniform)
    const uint32_t mg = (m16 >> (4 * g)) & 15u;                 // this group's
    a[0] = make_float4(wv[0] * a[0].x, wv[0] * a[0].y, wv[0] * a[0].z, wv[0] * a[0].w);
#pragma unroll
    for (int t = 1; t < 4; ++t) {
      const bool cut = (mg >> (t - 1)) & 1u;
      const float4 q = cut ? f4_zero() : a[t - 1];
      a[t] = f4_fma(wv[t], a[t], q);
    }
    // open tail of this group -> table (read only where the group's last slot ends no row); the tails a row of this group
    // continues from <- table.  Which groups end open is wave-uniform (m16): a closed group costs nothing below.
    tl[g][li] = a[3];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // entries below this group since the last row end below it: groups lo_h .. g - 1 (+ the carry when no row ended yet);
    // the item's open tail after this unit: groups lo_a .. G - 1
    const uint32_t below = m16 & ((1u << (4 * g)) - 1u);
    const int lo_h = below ? (31 - __builtin_clz(below)) >> 2 : 0;
    const int lo_a = m16 ? (31 - __builtin_clz(m16)) >> 2 : 0;
    float4 cin = below ? f4_zero() : carry;
    float4 cnext = m16 ? f4_zero() : carry;
#pragma unroll
    for (int h = 0; h < G; ++h) {
      if (!((m16 >> (4 * h + 3)) & 1u)) {                       // (uniform) group h ends inside a row
        const float4 th = tl[h][li];
        if (h < G - 1) cin = f4_add(cin, (h >= lo_h && h < g) ? th : f4_zero());
        if (h >= lo_a) cnext = f4_add(cnext, th);               // (uniform)
      }
    }
    carry = cnext;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // rows that end in this group: total at their last slot; the first one of the group takes the tails before it.  A trip
    // in which no group ends a row (uniform) issues its store out of range without forming a value.
    int row = p.rowbase + __builtin_popcount(below);
    bool first = true;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float4 o = a[t];
      uint32_t off = kOob;
      if (m16 & (0x11111111u & ((1u << S) - 1u)) << t) {        // (uniform)
        const bool e = (mg >> t) & 1u;
        o = f4_add(first ? f4_add(a[t], cin) : a[t], bv);
        off = e ? __umul24((uint32_t)row, pitch_y) + lo : kOob;
        if (SELF) {
          const uint32_t soff = e ? __umul24((uint32_t)row, pitch_x) + lo : kOob;
          o = f4_fma(self_coef, __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(sr, soff, 0, 0)), o);
        }
        if (e) {
          first = false;
          ++row;
        }
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4r, o), yr, off, 0, 0);
    }
    if (p.flags & F_LAST) {
      if (p.flags & F_EMPTY) {            // p.extra consecutive rows without entries: bias (+ self term) only
        for (int k = g; k < p.extra; k += G) {
          float4 o = bv;
          if (SELF) o = f4_fma(self_coef, *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(xs + (int64_t)(p.row0 + k) * ldx) + lo), o);
          *reinterpret_cast<float4*>(reinterpret_cast<char*>(y + (int64_t)(p.row0 + k) * ldy) + lo) = o;
        }
      }
      if (p.flags & F_MEMBER) {           // (block-uniform: the planner aligns the quadruples and the XCD ranges to 4)
        if (g == 0) grp_red[wave][li] = carry;
        __syncthreads();
        if (wave == 0 && g == 0) {
          float4 o = f4_add(f4_add(f4_add(grp_red[0][li], grp_red[1][li]), grp_red[2][li]), grp_red[3][li]);
          if (SELF) o = f4_fma(self_coef, *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(xs + (int64_t)p.row0 * ldx) + lo), o);
          o = f4_add(o, bv);
          *reinterpret_cast<float4*>(reinterpret_cast<char*>(y + (int64_t)p.row0 * ldy) + lo) = o;
        }
        __syncthreads();
      }
    }
  };

  float4 xa[4], xb[4];
  UnitP pa, pb;
  issue(xa, pa);
  // (four out-of-range stores - dropped by the buffer unit - so that the loop is entered with the memory-operation count it
  //  has on its back edge: the compiler takes the smaller of the two for the first wait of the body, which would then also
  //  wait for the previous unit's four stores on every trip)
#pragma unroll
  for (int t = 0; t < 4; ++t) __builtin_amdgcn_raw_buffer_store_b128(u32x4r{0u, 0u, 0u, 0u}, yr, kOob, 0, 0);
  for (;;) {                              // (the phases are pinned: left alone, the scheduler sums a unit BEFORE it requests the next)
    bool moved = false;
    if (u == nu) {
      advance();
      moved = true;
    }
    prefetch();
    issue(xb, pb);
    __builtin_amdgcn_sched_barrier(0);
    consume(xa, pa, moved);
    __builtin_amdgcn_sched_barrier(0);
    if (pa.flags & F_FINAL) break;
    moved = false;
    if (u == nu) {
      advance();
      moved = true;
    }
    prefetch();
    issue(xa, pa);
    __builtin_amdgcn_sched_barrier(0);
    consume(xb, pb, moved);
    __builtin_amdgcn_sched_barrier(0);
    if (pb.flags & F_FINAL) break;
  }
}

}  // namespace gd

extern "C" int gd_spmm_csr_runs_f32(const int32_t* items, int32_t n_items, const int32_t* col, const float* val, const float* x,
                                    int64_t ldx, float* y, int64_t ldy, const float* bias, float self_coef, const float* x_self,
                                    int32_t d, int32_t nnz, int32_t x_rows, const int32_t* xcd_bounds, void* stream) {
  using namespace gd;
  GD_REQUIRE(col && val && x && y && xcd_bounds && (items || n_items == 0), GD_E_NULL, "gd_spmm_csr_runs_f32: null pointer (val is required here)");
  GD_REQUIRE(n_items >= 0 && n_items % 4 == 0 && (d == 64 || d == 128) && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= d && ldy >= d, GD_E_DIM,
             "gd_spmm_csr_runs_f32: d must be 64 or 128 (got %d), 16-byte row strides, n_items a multiple of 4", d);
  GD_REQUIRE(aligned16(x) && aligned16(y) && aligned16(items) && (!bias || aligned16(bias)), GD_E_ALIGN,
             "gd_spmm_csr_runs_f32: unaligned pointer");
  GD_REQUIRE(x != y, GD_E_DIM, "gd_spmm_csr_runs_f32: x and y must not alias");
  GD_REQUIRE(x_rows > 0 && x_rows < (1 << 24) && ldx * 4 < (1 << 24) && ldy * 4 < (1 << 24) &&
                 (int64_t)x_rows * ldx * 4 < 0xfffffff0ll && (int64_t)x_rows * ldy * 4 < 0xfffffff0ll, GD_E_DIM,
             "gd_spmm_csr_runs_f32: x and y must be smaller than 4 GiB with row ids and pitches below 2^24 (use gd_spmm_csr_onepass_f32)");
  if (n_items == 0) return GD_OK;
  const float* xs = x_self ? x_self : x;
  GD_REQUIRE(aligned16(xs) && xs != y, GD_E_ALIGN, "gd_spmm_csr_runs_f32: bad x_self");
  int nblk = (n_items + 3) / 4;
  static const int cap = [] { const char* e = getenv("GD_SPMM_RUNS_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 8192; }();
  if (nblk > cap) nblk = cap;
  nblk = (nblk + 7) / 8 * 8;
  const uint32_t xb = (uint32_t)((int64_t)x_rows * ldx * 4), yb = (uint32_t)((int64_t)x_rows * ldy * 4);
  const int4* it = reinterpret_cast<const int4*>(items);
  hipStream_t s = (hipStream_t)stream;
#define GD_RUNS_LAUNCH(LPR, SELF)                                                                                          \
  hipLaunchKernelGGL((spmm_runs_kernel<LPR, SELF>), dim3(nblk), dim3(256), 0, s, it, xcd_bounds, col, val, x, ldx, y, ldy, bias, \
                     self_coef, xs, nnz, x_rows, xb, yb)
  if (d == 64) {
    if (self_coef != 0.0f) GD_RUNS_LAUNCH(16, true);
    else GD_RUNS_LAUNCH(16, false);
  } else {
    if (self_coef != 0.0f) GD_RUNS_LAUNCH(32, true);
    else GD_RUNS_LAUNCH(32, false);
  }
#undef GD_RUNS_LAUNCH
  return launched("spmm_runs");
}
