// Dense transform with an ARBITRARILY WIDE reduction dimension on the fp32 matrix cores:
//
//     out[r, :] = in[r, :] @ W (+ bias),   in [M, K] (K up to the bag-of-words widths: 1,639 / 8,710), W [K, N], N <= 128
//
// - the layer-1 transform x W1^T of the citation graphs (framework/models/gcn.py:11-12 through PyG's Linear), which
// the whole-weight-in-LDS row kernel (rows_gemm.hip) cannot take: its image is d_in * d_out * 4 bytes.
//
// Mapping to CDNA4: a block of 8 waves owns 256 rows (one 32-row tile per wave, operand rows straight from global
// memory, 64 contiguous bytes per lane and 32-wide k chunk, exactly as in rows_gemm.hip) and walks a k RANGE; the
// weight is streamed through LDS in 32-row chunks (16 KB at N = 128, double buffered, one barrier per chunk), in
// the tile-interleaved layout whose NT operands per k step are one 16-byte read.
// v_mfma_f32_32x32x2_f32: exact fp32 (a k-ordered fmaf chain per piece).
#include "common.h"

namespace gd {

using f32x16k = __attribute__((ext_vector_type(16))) float;
constexpr int kKtThreads = 512;

// Work partition ("stream-K"): the (row group, k chunk) units are numbered group-major and cut into equal ranges of
// `per` units, one range per block - a block walks its range and, whenever the range leaves a row group, parks the
// accumulators as one PIECE of that group.  Equal ranges instead of a groups x splits grid because M = 17,716 is 70 row
// groups: no integer number of k splits fills 256 CUs x 2 blocks evenly (5 splits = 350 blocks left a third of the
// chip idle half of the time).  Pieces are stored in the accumulators' own lane order (1 KB coalesced per store) and
// added in k order by the reduce kernel: deterministic, no atomics.
template <int NT>
__global__ __launch_bounds__(kKtThreads) void gemm_ktile_mfma_kernel(
    const float* __restrict__ in, int64_t ld_in, const int32_t* __restrict__ idx, int32_t n_rows,
    const float* __restrict__ w, int32_t n_chunks, int32_t per, int32_t n_units, const float* __restrict__ bias,
    float* __restrict__ out, int64_t ld_out, float* __restrict__ pieces, int32_t s_max) {
  constexpr int NTP = NT == 3 ? 4 : NT;
  constexpr int N = 32 * NT;
  constexpr int kChunkFloats = 32 * 32 * NTP;
  constexpr int PAIRS = (32 * 32) / kKtThreads;          // (k, r) pairs of a chunk per thread
  __shared__ __attribute__((aligned(16))) float wl[2][kChunkFloats];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r_lo = lane & 31, khalf = lane >> 5;
  const int u0 = blockIdx.x * per, u1 = min(n_units, u0 + per);
  if (u0 >= u1) return;
  const int n_tiles = (n_rows + 31) >> 5;

  auto operand_rows = [&](int g) -> const float4* {      // this lane's operand row of row group g
    const int s_a = min((g * 8 + wave) * 32 + r_lo, n_rows - 1);
    const int64_t row = idx ? idx[s_a] : s_a;
    return reinterpret_cast<const float4*>(in + row * ld_in) + khalf * 4;
  };
  float wreg[PAIRS][NTP];
  auto load_w = [&](int c) {
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      const int e = tid + p * kKtThreads, kk = e >> 5, r = e & 31;
      const float* wp = w + (int64_t)(c * 32 + kk) * N + r;
#pragma unroll
      for (int t = 0; t < NTP; ++t) wreg[p][t] = t < NT ? wp[32 * t] : 0.f;
    }
  };
  auto stash_w = [&](int buf) {
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      float* dst = wl[buf] + (tid + p * kKtThreads) * NTP;
      if (NTP == 4) *reinterpret_cast<float4*>(dst) = make_float4(wreg[p][0], wreg[p][1 % NTP], wreg[p][2 % NTP], wreg[p][3 % NTP]);
      else if (NTP == 2) *reinterpret_cast<float2*>(dst) = make_float2(wreg[p][0], wreg[p][1 % NTP]);
      else dst[0] = wreg[p][0];
    }
  };

  f32x16k acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  int g = u0 / n_chunks, c = u0 - g * n_chunks;
  const float4* src = operand_rows(g);
  float4 a_next[4];
  load_w(c);
#pragma unroll
  for (int i = 0; i < 4; ++i) a_next[i] = src[c * 8 + i];
  stash_w(0);
  __syncthreads();
  int cur = 0;
  for (int u = u0; u < u1; ++u) {
    float4 a4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a4[i] = a_next[i];
    const bool more = u + 1 < u1;
    const bool last_of_group = c + 1 == n_chunks;
    const int gn = last_of_group ? g + 1 : g, cn = last_of_group ? 0 : c + 1;
    const float4* srcn = src;
    if (more) {                                          // next unit's operands in flight behind this unit's MFMAs
      if (last_of_group) srcn = operand_rows(gn);
#pragma unroll
      for (int i = 0; i < 4; ++i) a_next[i] = srcn[cn * 8 + i];
      load_w(cn);
    }
    const float* wk = wl[cur] + (khalf * 16 * 32 + r_lo) * NTP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float av[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* wp = wk + (i * 4 + s) * 32 * NTP;
        float wv[NTP];
        if (NTP == 4) {
          const float4 f = *reinterpret_cast<const float4*>(wp);
          wv[0] = f.x; wv[1 % NTP] = f.y; wv[2 % NTP] = f.z; wv[3 % NTP] = f.w;
        } else if (NTP == 2) {
          const float2 f = *reinterpret_cast<const float2*>(wp);
          wv[0] = f.x; wv[1 % NTP] = f.y;
        } else {
          wv[0] = wp[0];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t], av[s], acc[t], 0, 0, 0);
      }
    }
    if (more) stash_w(cur ^ 1);
    __syncthreads();
    cur ^= 1;
    if (last_of_group || !more) {
      // D[i][j]: j = lane & 31 = sample, feature 32 t + 8 q + 4 khalf + c in acc[t][4 q + c] (as rows_gemm.hip)
      if (pieces) {                                      // piece number = blocks since the one holding the group's first unit
        const int piece = blockIdx.x - (g * n_chunks) / per;
        float4* dst = reinterpret_cast<float4*>(pieces + ((int64_t)g * s_max + piece) * (256 * N)) + wave * (NT * 4 * 64) + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            dst[(t * 4 + q) * 64] = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
      } else {
        const int tile = g * 8 + wave, s_a = tile * 32 + r_lo;
        if (tile < n_tiles && s_a < n_rows) {
          const int64_t orow = idx ? idx[s_a] : s_a;
          float* dst = out + orow * ld_out + 4 * khalf;
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float4 v = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
              if (bias) v = f4_add(v, *reinterpret_cast<const float4*>(bias + 32 * t + 8 * q + 4 * khalf));
              *reinterpret_cast<float4*>(dst + 32 * t + 8 * q) = v;
            }
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }
    g = gn; c = cn; src = srcn;
  }
}

// The same product with split arithmetic (gd_set_matrix_split(6), rows_gemm.hip: every fp32 product from six exact bf16
// partial products on v_mfma_f32_32x32x16_bf16, fp32 accumulation).  A 32-wide k chunk then lasts 0.7 us on the matrix
// cores instead of 2 us, too short to hide a memory round trip behind, so the pipeline is deeper: the unit of the stream-K
// partition is a MACRO chunk of 4 chunks (128 k) whose operand rows a lane keeps in registers (16 float4), the next
// unit's in flight in a second set; the weight streams through two 48 KB slots of two chunk images each (three bf16
// pieces in the instruction's operand order), cut from registers that were loaded one chunk pair earlier - one barrier
// per 64 k.
using bf16x8k = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2k = __attribute__((ext_vector_type(2))) __bf16;
using f32x2k = __attribute__((ext_vector_type(2))) float;
using u32x4k = __attribute__((ext_vector_type(4))) uint32_t;

__device__ inline void split8k(const float (&v)[8], bf16x8k (&s)[3]) {
  u32x4k p[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x2k r = {v[2 * i], v[2 * i + 1]};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const uint32_t w = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2k));
      p[q][i] = w;
      if (q < 2) {
        f32x2k h;
        h[0] = __builtin_bit_cast(float, w << 16);
        h[1] = __builtin_bit_cast(float, w & 0xffff0000u);
        r = r - h;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) s[q] = __builtin_bit_cast(bf16x8k, p[q]);
}

// IDX: rows through an index list (compile-time: a load under a run-time branch makes the count of loads in flight
// path-dependent, and every later wait becomes s_waitcnt vmcnt(0) - which drains the operand prefetch)
template <int NT, bool IDX>
__global__ __launch_bounds__(kKtThreads, 2) void gemm_ktile_split_kernel(
    const float* __restrict__ in, int64_t ld_in, const int32_t* __restrict__ idx, int32_t n_rows,
    const float* __restrict__ w, int32_t n_chunks, int32_t n_mac, int32_t per, int32_t n_units, const float* __restrict__ bias,
    float* __restrict__ out, int64_t ld_out, float* __restrict__ pieces, int32_t s_max) {
  constexpr int N = 32 * NT;
  constexpr int kOps = 2 * NT * 2 * 32;                  // 16-byte operands of one piece of one chunk: [m][t][khalf][r]
  extern __shared__ __attribute__((aligned(16))) unsigned char kt_lds[];
  bf16x8k* const wimg = reinterpret_cast<bf16x8k*>(kt_lds);     // [slot 2][chunk of the pair 2][piece 3][kOps]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r_lo = lane & 31, khalf = lane >> 5;
  const int u0 = blockIdx.x * per, u1 = min(n_units, u0 + per);
  if (u0 >= u1) return;
  const int n_tiles = (n_rows + 31) >> 5;

  // ---- weight stream: position = (unit, chunk j of its macro chunk).  Every load below is UNCONDITIONAL (clamped
  // addresses): a load under a branch makes the number of loads in flight path-dependent, and the compiler then waits
  // for ALL of them (s_waitcnt vmcnt(0)) in front of every chunk - the operand prefetch and the weight chunk requested
  // a moment ago included (measured: 1.8 us of stall per 0.7 us chunk)
  int wu = u0, wj = 0;                                   // position of the next chunk PAIR to load (wj = 0 or 2)
  bool w_pending = false;                                // wreg holds a pair that is not in LDS yet
  float wreg[2][8];
  const bool w_thread = tid < 4 * N;
  const int w_kg = min(tid / N, 3), w_n = tid % N;       // this thread's (8-k group, feature) item of every chunk
  auto load_w = [&]() {                                  // wreg <- the two chunks at (wu, wj), (wu, wj + 1); advance
    const bool valid = wu < u1;
    const int uu = valid ? wu : u1 - 1, jj = valid ? wj : 2;
    const int cg = (uu % n_mac) * 4 + jj;
    const float* wp = w + (int64_t)(cg * 32 + 8 * w_kg) * N + w_n;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < 8; ++c) wreg[h][c] = wp[(h * 32 + c) * N];
    w_pending = valid;
    if (valid) { wj += 2; if (wj == 4) { wj = 0; ++wu; } }
  };
  auto stash_w = [&](int slot) {
    if (w_pending && w_thread) {
      const int off = (((w_kg & 1) * NT + (w_n >> 5)) * 2 + ((w_kg >> 1) & 1)) * 32 + (w_n & 31);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bf16x8k pc[3];
        split8k(wreg[h], pc);
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) wimg[((slot * 2 + h) * 3 + qq) * kOps + off] = pc[qq];
      }
    }
  };

  // ---- operand rows: the unit's 4 x 64 bytes of this lane's row (its khalf of every chunk)
  auto fetch = [&](int u_, float4 (&a)[16]) {
    const int u = min(u_, u1 - 1);
    const int g = u / n_mac, cm = u - g * n_mac;
    const int s_a = min((g * 8 + wave) * 32 + r_lo, n_rows - 1);
    const int64_t row = IDX ? idx[s_a] : s_a;
    const float4* src = reinterpret_cast<const float4*>(in + row * ld_in) + khalf * 4 + (cm * 4) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) a[j * 4 + i] = src[j * 8 + i];
  };

  f32x16k acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  int q = 0;                                             // chunks done by this block: chunk q is chunk q & 1 of slot (q >> 1) & 1
  auto work = [&](int u, float4 (&a)[16]) {
    const int g = u / n_mac, cm = u - g * n_mac;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      {
        const bf16x8k* wk = wimg + (((q >> 1) & 1) * 2 + (j & 1)) * 3 * kOps + khalf * 32 + r_lo;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const float4 lo = a[j * 4 + 2 * m], hi = a[j * 4 + 2 * m + 1];
          const float v8[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          bf16x8k sp[3];
          split8k(v8, sp);
          // two output tiles at a time: consecutive matrix instructions accumulate into DIFFERENT registers, so the
          // wave issues them back to back instead of waiting out each one's latency on a six-deep dependent chain
#pragma unroll
          for (int t0 = 0; t0 < NT; t0 += 2) {
            constexpr int kPairs[6][2] = {{2, 0}, {1, 1}, {0, 2}, {1, 0}, {0, 1}, {0, 0}};    // (weight piece, sample piece), small terms first
            bf16x8k wq[2][3];
#pragma unroll
            for (int d = 0; d < 2; ++d)
              if (t0 + d < NT) {
                const bf16x8k* wo = wk + (m * NT + t0 + d) * 64;
                wq[d][0] = wo[0]; wq[d][1] = wo[kOps]; wq[d][2] = wo[2 * kOps];
              }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
              for (int d = 0; d < 2; ++d)
                if (t0 + d < NT)
                  acc[t0 + d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[d][kPairs[pr][0]], sp[kPairs[pr][1]], acc[t0 + d], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (j & 1) {                                     // one barrier per PAIR of chunks (64 k): the waves of a block leave
          stash_w(((q >> 1) + 1) & 1);                   // it in step, so whatever is not matrix work between two barriers
          load_w();                                      // idles the matrix cores of the whole CU - half as often this way
          __syncthreads();
        }
        ++q;
      }
    }
    if (cm + 1 == n_mac || u + 1 == u1) {
      // D[i][j]: j = lane & 31 = sample, feature 32 t + 8 q + 4 khalf + c in acc[t][4 q + c] (as rows_gemm.hip)
      if (pieces) {                                      // piece number = blocks since the one holding the group's first unit
        const int piece = blockIdx.x - (g * n_mac) / per;
        float4* dst = reinterpret_cast<float4*>(pieces + ((int64_t)g * s_max + piece) * (256 * N)) + wave * (NT * 4 * 64) + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)
            dst[(t * 4 + qq) * 64] = make_float4(acc[t][4 * qq], acc[t][4 * qq + 1], acc[t][4 * qq + 2], acc[t][4 * qq + 3]);
      } else {
        const int tile = g * 8 + wave, s_a = tile * 32 + r_lo;
        if (tile < n_tiles && s_a < n_rows) {
          const int64_t orow = IDX ? idx[s_a] : s_a;
          float* dst = out + orow * ld_out + 4 * khalf;
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              float4 v = make_float4(acc[t][4 * qq], acc[t][4 * qq + 1], acc[t][4 * qq + 2], acc[t][4 * qq + 3]);
              if (bias) v = f4_add(v, *reinterpret_cast<const float4*>(bias + 32 * t + 8 * qq + 4 * khalf));
              *reinterpret_cast<float4*>(dst + 32 * t + 8 * qq) = v;
            }
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }
  };

  float4 ra[16], rb[16];
  fetch(u0, ra);
  load_w(); stash_w(0);
  load_w();
  __syncthreads();
  for (int u = u0; u < u1; u += 2) {
    fetch(u + 1, rb);
    work(u, ra);
    if (u + 1 >= u1) break;
    fetch(u + 2, ra);
    work(u + 1, rb);
  }
}

// out[row, :] = bias + the pieces of the row's group in k order.  One thread per float4 of the piece layout
// (coalesced reads of every piece; the 16-byte results go to their rows).
__global__ __launch_bounds__(256) void gemm_ktile_reduce_kernel(const float* __restrict__ pieces, int32_t s_max, int32_t n_chunks,
                                                                int32_t per, int32_t nt, const int32_t* __restrict__ idx,
                                                                int32_t n_rows, const float* __restrict__ bias,
                                                                float* __restrict__ out, int64_t ld_out) {
  const int per_group = 8 * nt * 4 * 64;                 // float4s of one piece
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int g = (int)(e / per_group), v = (int)(e % per_group);
  const int lane = v & 63, tq = (v >> 6) % (nt * 4), wave = v / (nt * 4 * 64);
  const int s_a = (g * 8 + wave) * 32 + (lane & 31);
  if (s_a >= n_rows) return;
  const int col = 32 * (tq >> 2) + 8 * (tq & 3) + 4 * (lane >> 5);
  const int p_first = (g * n_chunks) / per, p_last = ((g + 1) * n_chunks - 1) / per;
  const float4* p = reinterpret_cast<const float4*>(pieces + (int64_t)g * s_max * (256 * 32 * nt)) + v;
  float4 acc = bias ? *reinterpret_cast<const float4*>(bias + col) : f4_zero();
  const int n_p = p_last - p_first + 1;
  int k = 0;
  for (; k + 2 <= n_p; k += 2) {
    const float4 a = p[(int64_t)k * per_group], b = p[(int64_t)(k + 1) * per_group];
    acc = f4_add(f4_add(acc, a), b);
  }
  if (k < n_p) acc = f4_add(acc, p[(int64_t)k * per_group]);
  const int64_t row = idx ? idx[s_a] : s_a;
  *reinterpret_cast<float4*>(out + row * ld_out + col) = acc;
}

struct KtileGeometry { int groups, chunks, per, blocks, s_max; };   // s_max == 0: whole groups per block, no pieces

static KtileGeometry ktile_geometry(int32_t n_rows, int32_t k) {
  KtileGeometry q;
  q.groups = (n_rows + 255) / 256;
  q.chunks = k / 32;
  const int64_t units = (int64_t)q.groups * q.chunks;
  // 2 blocks per CU (16 waves) sustain the matrix cores best; when that leaves fewer than 12 chunks behind every
  // 128 KB piece (M = 17,716, K = 1,664: 8), one block per CU halves the piece traffic and wins (88 vs 93 us)
  static const int forced = [] { const char* e = getenv("GD_KTILE_SLOTS"); return e ? atoi(e) : 0; }();
  int per = (int)((units + 511) / 512);
  if (per < 12) per = (int)((units + 255) / 256);
  if (forced > 0) per = (int)((units + forced - 1) / forced);
  if (per < 4) per = 4;                                  // a piece costs a 128 KB round trip: at least 4 chunks of work behind it
  if (per >= q.chunks) {
    per = (per + q.chunks - 1) / q.chunks * q.chunks;
    q.s_max = 0;
  } else {
    q.s_max = (q.chunks + per - 2) / per + 1;
  }
  q.per = per;
  q.blocks = (int)((units + per - 1) / per);
  return q;
}

// split form: units are macro chunks (128 k), one block per CU (72 KB of LDS, 200+ registers)
static KtileGeometry ktile_geometry_split(int32_t n_rows, int32_t k) {
  KtileGeometry q;
  q.groups = (n_rows + 255) / 256;
  q.chunks = k / 128;                                    // macro chunks (k % 128 == 0)
  const int64_t units = (int64_t)q.groups * q.chunks;
  int per = (int)((units + 255) / 256);
  if (per < 1) per = 1;
  if (per >= q.chunks) {
    per = (per + q.chunks - 1) / q.chunks * q.chunks;
    q.s_max = 0;
  } else {
    q.s_max = (q.chunks + per - 2) / per + 1;
  }
  q.per = per;
  q.blocks = (int)((units + per - 1) / per);
  return q;
}

}  // namespace gd

extern "C" int64_t gd_gemm_f32_workspace(int32_t n_rows, int32_t k, int32_t n) {
  if (n_rows <= 0 || k <= 0 || k % 32) return 0;
  // (the larger of the two forms: the arithmetic switch may change between this call and the product)
  const gd::KtileGeometry q = gd::ktile_geometry(n_rows, k);
  const int s_split = k % 128 == 0 ? gd::ktile_geometry_split(n_rows, k).s_max : 0;
  return (int64_t)q.groups * (q.s_max > s_split ? q.s_max : s_split) * 256 * n;
}

extern "C" int gd_gemm_f32(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_rows, const float* w, int32_t k,
                           int32_t n, const float* bias, float* out, int64_t ld_out, float* workspace, void* stream) {
  using namespace gd;
  GD_REQUIRE(in && w && out, GD_E_NULL, "gd_gemm_f32: null pointer");
  GD_REQUIRE(n_rows >= 0 && k > 0 && k % 32 == 0 && n % 32 == 0 && n >= 32 && n <= 128 && ld_in >= k && ld_out >= n &&
                 ld_in % 4 == 0 && ld_out % 4 == 0, GD_E_DIM,
             "gd_gemm_f32: needs k %% 32 == 0 (pad with zeros), n in {32, 64, 96, 128}, 16-byte row pitches (k=%d n=%d)", k, n);
  GD_REQUIRE(aligned16(in) && aligned16(out) && aligned16(w) && (!bias || aligned16(bias)) && in != out, GD_E_ALIGN,
             "gd_gemm_f32: unaligned or aliasing pointer");
  if (n_rows == 0) return GD_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool split = matrix_split() == 6 && k % 128 == 0;     // (other widths: the fp32 instruction; ops.gemm_wide pads)
  const KtileGeometry q = split ? ktile_geometry_split(n_rows, k) : ktile_geometry(n_rows, k);
  GD_REQUIRE(q.s_max == 0 || (workspace && aligned16(workspace)), GD_E_NULL, "gd_gemm_f32: workspace of gd_gemm_f32_workspace() floats");
  const dim3 grid(q.blocks), block(kKtThreads);
  float* pieces = q.s_max ? workspace : nullptr;
  if (split) {
    const size_t lds = (size_t)2 * 2 * 3 * (2 * (n / 32) * 2 * 32) * 16;        // two slots of two chunks of three pieces
#define GD_KS_CASE(NT)                                                                                                        \
  do {                                                                                                                        \
    auto kern = idx ? gemm_ktile_split_kernel<NT, true> : gemm_ktile_split_kernel<NT, false>;                                 \
    /* the LDS limit of both instantiations is raised once per process, not on every launch */                               \
    static const hipError_t once = [] {                                                                                       \
      const hipError_t a = hipFuncSetAttribute((const void*)gemm_ktile_split_kernel<NT, true>,                                \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);                       \
      const hipError_t b = hipFuncSetAttribute((const void*)gemm_ktile_split_kernel<NT, false>,                               \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);                       \
      return a != hipSuccess ? a : b;                                                                                         \
    }();                                                                                                                      \
    GD_REQUIRE(once == hipSuccess, -(int)once, "gd_gemm_f32: cannot raise the LDS limit");                                    \
    hipLaunchKernelGGL(kern, grid, block, lds, s, in, ld_in, idx, n_rows, w, k / 32, q.chunks, q.per, q.groups * q.chunks,     \
                       bias, out, ld_out, pieces, q.s_max);                                                                   \
  } while (0)
    switch (n / 32) {
      case 1: GD_KS_CASE(1); break;
      case 2: GD_KS_CASE(2); break;
      case 3: GD_KS_CASE(3); break;
      default: GD_KS_CASE(4); break;
    }
#undef GD_KS_CASE
    int rc = launched("gemm_ktile_split");
    if (rc || !q.s_max) return rc;
    const int64_t total = (int64_t)q.groups * 8 * (n / 32) * 4 * 64;
    hipLaunchKernelGGL(gemm_ktile_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, workspace, q.s_max, q.chunks,
                       q.per, n / 32, idx, n_rows, bias, out, ld_out);
    return launched("gemm_ktile_reduce");
  }
#define GD_KT_CASE(NT) \
  hipLaunchKernelGGL((gemm_ktile_mfma_kernel<NT>), grid, block, 0, s, in, ld_in, idx, n_rows, w, q.chunks, q.per, q.groups * q.chunks, \
                     bias, out, ld_out, pieces, q.s_max)
  switch (n / 32) {
    case 1: GD_KT_CASE(1); break;
    case 2: GD_KT_CASE(2); break;
    case 3: GD_KT_CASE(3); break;
    default: GD_KT_CASE(4); break;
  }
#undef GD_KT_CASE
  int rc = launched("gemm_ktile");
  if (rc || !q.s_max) return rc;
  const int64_t total = (int64_t)q.groups * 8 * (n / 32) * 4 * 64;
  hipLaunchKernelGGL(gemm_ktile_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, workspace, q.s_max, q.chunks, q.per,
                     n / 32, idx, n_rows, bias, out, ld_out);
  return launched("gemm_ktile_reduce");
}
