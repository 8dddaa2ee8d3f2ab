// Lab (round 4): weight-stationary register form of the row GEMM  out[idx[s], :] = in[idx[s], :] @ W   (W = [DIN][32 NT]).
//
// Production form (csrc/rows_gemm.hip): the weight operand of every v_mfma_f32_32x32x2_f32 comes from a 64 KB LDS image,
// four waves per SIMD; knock-outs (NOTES, rounds 1-3) put 25 % of the time on that LDS operand.  Here ONE wave per SIMD keeps
// the whole weight as MFMA A-fragments in its registers (128 x 128 = 256 registers per lane), the sample rows of the current
// 32-row tile and of the NEXT one in 2 x DIN/2 registers, 16 NT accumulators: no LDS, no barrier, nothing but matrix
// instructions between a tile's loads and its stores.
//   lane (r = lane & 31, h = lane >> 5) holds x[row r][h KH + s], s = 0 .. KH-1 (KH = DIN / 2; contiguous: 16-byte loads)
//   step s multiplies A = W[h KH + s][32 t + r]  (register wr[t][s])  with  B = x[r][h KH + s]   (the k order is permuted,
//   both operands agree), D[i][j]: j = sample, i = output feature - every lane ends up with 4-float runs of its own row.
//
//   hipcc -O3 --offload-arch=gfx950 ws_gemm_lab.hip -o ws_gemm_lab.bin ;  ./ws_gemm_lab.bin [rows] [reps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ORDER 0: rotate over all NT accumulators inside each k step (stores of a tile all at its end)
// ORDER 1: output tiles in pairs - pair (0,1) over all k, its stores, then pair (2,3): the stores of a pair issue while the
//          other pair's matrix instructions run
// HALF: work unit = (row tile, half of the output tiles) so that the units divide evenly over the waves (tail balance)
template <int NT, int DIN, int ORDER, bool SIGNS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ws_gemm_kernel(
    const float* __restrict__ in, int64_t ld_in, const int32_t* __restrict__ idx, int n_sel, const float* __restrict__ w,
    float* __restrict__ out, int64_t ld_out, uint32_t* __restrict__ sign_out) {
  constexpr int KH = DIN / 2, DOUT = 32 * NT, XV = KH / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n_tiles = (n_sel + 31) >> 5;
  const int n_waves = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
  // contiguous tile range per wave (rows of a wave are neighbours: the index list is sorted)
  const int t_lo = (int)((int64_t)n_tiles * wid / n_waves), t_hi = (int)((int64_t)n_tiles * (wid + 1) / n_waves);
  if (t_lo >= t_hi) return;

  float wr[NT][KH];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < KH; ++s) wr[t][s] = w[(int64_t)(h * KH + s) * DOUT + 32 * t + r];

  auto row_of = [&](int tile) -> int32_t {
    const int s_ = min(tile * 32 + r, n_sel - 1);
    return idx ? idx[s_] : s_;
  };
  auto fetch = [&](int32_t row, float4 (&x)[XV]) {
    const float4* src = reinterpret_cast<const float4*>(in + (int64_t)row * ld_in + h * KH);
#pragma unroll
    for (int i = 0; i < XV; ++i) x[i] = src[i];
  };
  auto store_tile = [&](const f32x16& a, int t, float* dst, uint32_t& pos) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
      if (SIGNS)
        pos |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << (8 * q + 4 * h);
      *reinterpret_cast<float4*>(dst + 32 * t + 8 * q) = v;
    }
  };
  auto work = [&](int tile, int32_t row, const float4 (&x)[XV]) {
    const int s_a = tile * 32 + r;
    const bool live = s_a < n_sel;
    float* dst = out + (int64_t)row * ld_out + 4 * h;
    if (!live) dst = out + (int64_t)row * ld_out + 4 * h;      // clamped row: rewrites the last row with the same values
    if (ORDER == 0) {
      f32x16 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const float xv[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[t][4 * i + c], xv[c], acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        uint32_t pos = 0;
        store_tile(acc[t], t, dst, pos);
        if (SIGNS) {
          pos |= (uint32_t)__shfl_xor((int)pos, 32);
          if (h == 0 && live) sign_out[(int64_t)s_a * NT + t] = pos;
        }
      }
    } else {
#pragma unroll
      for (int t0 = 0; t0 < NT; t0 += 2) {
        f32x16 acc[2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[d][i] = 0.f;
#pragma unroll
        for (int i = 0; i < XV; ++i) {
          const float xv[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int d = 0; d < 2; ++d)
              if (t0 + d < NT) acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[t0 + d][4 * i + c], xv[c], acc[d], 0, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < 2; ++d)
          if (t0 + d < NT) {
            uint32_t pos = 0;
            store_tile(acc[d], t0 + d, dst, pos);
            if (SIGNS) {
              pos |= (uint32_t)__shfl_xor((int)pos, 32);
              if (h == 0 && live) sign_out[(int64_t)s_a * NT + t0 + d] = pos;
            }
          }
      }
    }
  };

  // Row ids are requested two tiles before the rows they address (an index -> row chain in front of a tile would be two
  // exposed round trips with one wave per SIMD); the rows of tile T+1 are in flight while tile T multiplies.  One tile per
  // loop trip, the in-flight rows handed over by register copies (64 moves under 256 matrix instructions): with a
  // two-tile body and an exit test between the halves the compiler sinks the second half's loads behind the test.
  float4 xc[XV], xn[XV];
  int32_t r_cur, r_nxt = row_of(t_lo), r_nn = row_of(min(t_lo + 1, n_tiles - 1));
  fetch(r_nxt, xn);
  for (int tile = t_lo; tile < t_hi; ++tile) {
#pragma unroll
    for (int i = 0; i < XV; ++i) xc[i] = xn[i];
    r_cur = r_nxt;
    r_nxt = r_nn;
    r_nn = row_of(min(tile + 2, n_tiles - 1));
    fetch(r_nxt, xn);
    __builtin_amdgcn_sched_barrier(0);       // the next tile's loads stay IN FRONT of this tile's matrix instructions
    work(tile, r_cur, xc);
  }
}

// Ring form: ONE set of row registers.  The k steps of a tile run in NCH chunks; as soon as a chunk's matrix instructions are
// issued its registers are reloaded with the SAME chunk of the next tile, so every load has (NCH - 1) / NCH of a tile to
// land, nothing is copied and 64 registers are free again.  sched_barriers pin the order (the scheduler otherwise moves the
// loads to where their values are used, i.e. in front of the wait).
// KO (timing only): bit 0 = no output stores, bit 1 = no row loads in the loop (the first tile's rows are re-used)
template <int NT, int DIN, int NCH, bool SIGNS, bool HASIDX, int KO = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ws_ring_kernel(
    const float* __restrict__ in, int64_t ld_in, const int32_t* __restrict__ idx, int n_sel, const float* __restrict__ w,
    float* __restrict__ out, int64_t ld_out, uint32_t* __restrict__ sign_out, unsigned long long* __restrict__ stamps) {
  constexpr int KH = DIN / 2, DOUT = 32 * NT, XV = KH / 4, CV = XV / NCH;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n_tiles = (n_sel + 31) >> 5;
  const int n_waves = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
  const int t_lo = (int)((int64_t)n_tiles * wid / n_waves), t_hi = (int)((int64_t)n_tiles * (wid + 1) / n_waves);
  if (t_lo >= t_hi) return;
  float wr[NT][KH];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < KH; ++s) wr[t][s] = w[(int64_t)(h * KH + s) * DOUT + 32 * t + r];
  auto row_of = [&](int tile) -> int32_t {
    const int s_ = min(tile * 32 + r, n_sel - 1);
    return HASIDX ? idx[s_] : s_;       // (a load under a branch makes every wait of the loop a vmcnt(0))
  };
  float4 x[XV];
  int32_t r_cur = row_of(t_lo), r_nxt = row_of(min(t_lo + 1, n_tiles - 1));
  {
    const float4* src = reinterpret_cast<const float4*>(in + (int64_t)r_cur * ld_in + h * KH);
#pragma unroll
    for (int i = 0; i < XV; ++i) x[i] = src[i];
  }
  for (int tile = t_lo; tile < t_hi; ++tile) {
    const int32_t r_nn = row_of(min(tile + 2, n_tiles - 1));       // used a tile from now
    const float4* nsrc = reinterpret_cast<const float4*>(in + (int64_t)r_nxt * ld_in + h * KH);
    const int s_a = tile * 32 + r;
    const bool live = s_a < n_sel;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int i = c * CV; i < (c + 1) * CV; ++i) {
        const float xv[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[t][4 * i + q], xv[q], acc[t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = c * CV; i < (c + 1) * CV; ++i)
        if (!(KO & 2) || n_sel < 0) x[i] = nsrc[i];
      __builtin_amdgcn_sched_barrier(0);
    }
    float* dst = out + (int64_t)r_cur * ld_out + 4 * h;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      uint32_t pos = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
        if (SIGNS)
          pos |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << (8 * q + 4 * h);
        if (!(KO & 1) || n_sel < 0) *reinterpret_cast<float4*>(dst + 32 * t + 8 * q) = v;
      }
      if (SIGNS) {
        pos |= (uint32_t)__shfl_xor((int)pos, 32);
        if (h == 0 && live) sign_out[(int64_t)s_a * NT + t] = pos;
      }
    }
    r_cur = r_nxt;
    r_nxt = r_nn;
  }
  if (stamps && lane == 0) {       // shader cycles and 100 MHz wall ticks this wave was alive, tiles it processed
    stamps[3 * wid + 0] = __builtin_readcyclecounter() - c0;
    stamps[3 * wid + 1] = wall_clock64() - w0;
    stamps[3 * wid + 2] = t_hi - t_lo;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 16-row form (v_mfma_f32_16x16x4_f32, 32 cycles per instruction on a SIMD, the same 64 FLOP/clk): work units of 16 rows divide
// twice as finely over the 1024 waves (tail), a store instruction writes 64 contiguous bytes of 16 rows (32 bytes of 32 before),
// a row tile is 32 registers and its accumulators 4 per output tile.  lane (r = lane & 15, kq = lane >> 4) holds
// x[row r][kq KQ + s], s = 0 .. KQ-1 (KQ = DIN / 4: 16-byte loads), step s multiplies A = W[kq KQ + s][16 t + r] (register
// wr[t][s]) with it; D: lane (r, kq) ends with outputs 16 t + 4 kq + c of its row.
// The weight reaches the registers through LDS (one coalesced pass over W per block instead of 256 scalar loads per lane).
// The output tiles run in two groups: while group 1's matrix instructions issue, group 0's results are stored and the row
// registers already consumed are reloaded with the next unit's rows; group 1's results are stored under group 0 of the NEXT unit.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int DIN, int DOUT, bool HASIDX, bool SIGNS, int KO = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void ws16_kernel(
    const float* __restrict__ in, int64_t ld_in, const int32_t* __restrict__ idx, int n_sel, const float* __restrict__ w,
    float* __restrict__ out, int64_t ld_out, uint32_t* __restrict__ sign_out, unsigned long long* __restrict__ stamps) {
  constexpr int KQ = DIN / 4, NT = DOUT / 16, XV = KQ / 4, NG = NT / 2, NW = DOUT / 32;
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar loop control)
  const int r = lane & 15, kq = lane >> 4;
  // weight image: wl[k DOUT + n + 16 (k / KQ)] - the two k quarters a 32-lane LDS access touches sit 16 banks apart
  for (int e = tid; e < DIN * DOUT / 4; e += 256) {
    const int k = e / (DOUT / 4), n4 = e % (DOUT / 4);
    *reinterpret_cast<float4*>(wl + k * DOUT + 4 * n4 + 16 * (k / KQ)) = reinterpret_cast<const float4*>(w)[e];
  }
  __syncthreads();
  const unsigned long long cA = __builtin_readcyclecounter();
  float wr[NT][KQ];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < KQ; ++s) wr[t][s] = wl[(kq * KQ + s) * DOUT + 16 * t + r + 16 * kq];

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long c1 = __builtin_readcyclecounter();
  const int n_units = (n_sel + 15) >> 4;
  const int n_waves = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
  const int u_lo = (int)((int64_t)n_units * wid / n_waves), u_hi = (int)((int64_t)n_units * (wid + 1) / n_waves);
  if (u_lo < u_hi) {
    // No branch inside the loop (a load or store under a branch turns every wait of the loop into vmcnt(0)): rows past the
    // end are clamped to the last row - their lanes recompute and rewrite that row with identical values.
    auto row_of = [&](int u) -> int32_t {
      const int s_ = min(u * 16 + r, n_sel - 1);
      return HASIDX ? idx[s_] : s_;
    };
    float4 x[XV];
    int32_t r_cur = row_of(u_lo), r_nxt = row_of(min(u_lo + 1, n_units - 1));
    {
      const float4* src = reinterpret_cast<const float4*>(in + (int64_t)r_cur * ld_in + kq * KQ);
#pragma unroll
      for (int i = 0; i < XV; ++i) x[i] = src[i];
    }
    // shadow of the previous unit's results: stored one tile per slot while this unit's matrix instructions issue.
    // First trip: zeros to the first unit's own rows, overwritten by its results a trip later (same wave, same
    // addresses: program order).
    f32x4v sh[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) sh[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float* dst_prev = out + (int64_t)r_cur * ld_out + 4 * kq;
    int sa_prev = min(u_lo * 16 + r, n_sel - 1);
    uint32_t sg[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) sg[q] = 0;
    auto store_tile = [&](int t) {
      const float4 v = make_float4(sh[t][0], sh[t][1], sh[t][2], sh[t][3]);
      if (SIGNS) {
        const uint32_t b = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
        sg[t >> 1] |= b << (16 * (t & 1) + 4 * kq);
      }
      if (!(KO & 1) || n_sel < 0) *reinterpret_cast<float4*>(dst_prev + 16 * t) = v;
    };
    auto store_signs = [&]() {          // the four lanes of a row merge their bits; lane kq writes word kq (mod NW)
      uint32_t mine = 0;
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        uint32_t v = sg[q];
        v |= (uint32_t)__shfl_xor((int)v, 16);
        v |= (uint32_t)__shfl_xor((int)v, 32);
        if ((kq & (NW - 1)) == q) mine = v;
        sg[q] = 0;
      }
      sign_out[(int64_t)sa_prev * NW + (kq & (NW - 1))] = mine;
    };
    for (int u = u_lo; u < u_hi; ++u) {
      const int32_t r_nn = row_of(min(u + 2, n_units - 1));       // used a unit from now
      const float4* nsrc = reinterpret_cast<const float4*>(in + (int64_t)r_nxt * ld_in + kq * KQ);
      f32x4v acc[NT];
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const float xv[4] = {x[i].x, x[i].y, x[i].z, x[i].w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (i == 0 && c == 0) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t][0], xv[0], f32x4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t][4 * i + c], xv[c], acc[t], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
        if (!(KO & 2) || n_sel < 0) x[i] = nsrc[i];                // the registers just consumed: same chunk of the next unit
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (t * XV / NT == i) store_tile(t);
        if (SIGNS && i == XV - 1) store_signs();
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) sh[t] = acc[t];
      dst_prev = out + (int64_t)r_cur * ld_out + 4 * kq;
      sa_prev = min(u * 16 + r, n_sel - 1);
      r_cur = r_nxt;
      r_nxt = r_nn;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) store_tile(t);
    if (SIGNS) store_signs();
  }
  if (stamps && lane == 0) {
    stamps[3 * wid + 0] = __builtin_readcyclecounter() - c0;
    stamps[3 * wid + 1] = wall_clock64() - w0;
    stamps[3 * wid + 2] = (unsigned long long)(u_hi - u_lo) | ((c1 - c0) << 32) | ((cA - c0) << 48 >> 32 << 16 >> 16 << 0) * 0;
    stamps[3 * 1024 + wid] = cA - c0;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
using kern_t = void (*)(const float*, int64_t, const int32_t*, int, const float*, float*, int64_t, uint32_t*);
template <int NT, int DIN>
static float run_k(kern_t k, const char* name, const float* in, const int32_t* idx, int n_sel, const float* w, float* out, uint32_t* signs,
                 int reps, int grid) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, (int64_t)DIN, idx, n_sel, w, out, (int64_t)(32 * NT), signs);
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0.f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, (int64_t)DIN, idx, n_sel, w, out, (int64_t)(32 * NT), signs);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms);
    sum += ms;
  }
  const double gf = 2.0 * n_sel * DIN * 32 * NT * 1e-9;
  printf("%-34s rows %7d  %3d->%3d  grid %4d : best %7.1f us  mean %7.1f us  = %6.1f TF (best)\n", name, n_sel, DIN, 32 * NT, grid,
         best * 1e3, sum / reps * 1e3, gf / best);
  return best;
}

template <int NT, int DIN, int ORDER, bool SIGNS>
static float run(const char* name, const float* in, const int32_t* idx, int n_sel, const float* w, float* out, uint32_t* signs,
                 int reps, int grid) {
  return run_k<NT, DIN>(ws_gemm_kernel<NT, DIN, ORDER, SIGNS>, name, in, idx, n_sel, w, out, signs, reps, grid);
}
static unsigned long long* g_stamps = nullptr;
template <int NT, int DIN, int NCH, bool SIGNS, int KO = 0>
static float run_ring(const char* name, const float* in, const int32_t* idx, int n_sel, const float* w, float* out, uint32_t* signs,
                      int reps, int grid) {
  if (!g_stamps) CK(hipMalloc(&g_stamps, 8 * 8 * 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (idx) hipLaunchKernelGGL((ws_ring_kernel<NT, DIN, NCH, SIGNS, true, KO>), dim3(grid), dim3(256), 0, 0, in, (int64_t)DIN, idx, n_sel, w, out, (int64_t)(32 * NT), signs, g_stamps);
    else hipLaunchKernelGGL((ws_ring_kernel<NT, DIN, NCH, SIGNS, false, KO>), dim3(grid), dim3(256), 0, 0, in, (int64_t)DIN, idx, n_sel, w, out, (int64_t)(32 * NT), signs, g_stamps);
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0.f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms);
    sum += ms;
  }
  std::vector<unsigned long long> st(3 * grid * 4);
  CK(hipMemcpy(st.data(), g_stamps, st.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, wall = 0, tiles = 0, wmax = 0, wmin = 1e18;
  for (int i = 0; i < grid * 4; ++i) {
    cyc += st[3 * i]; wall += st[3 * i + 1]; tiles += st[3 * i + 2];
    wmax = fmax(wmax, (double)st[3 * i + 1]); wmin = fmin(wmin, (double)st[3 * i + 1]);
  }
  const double gf = 2.0 * n_sel * DIN * 32 * NT * 1e-9;
  printf("%-34s rows %7d  %3d->%3d  KO %d : best %7.1f us  mean %7.1f us  = %6.1f TF | clock %.2f GHz, %.0f cycles/tile, wave life %.1f .. %.1f us\n",
         name, n_sel, DIN, 32 * NT, KO, best * 1e3, sum / reps * 1e3, gf / best, cyc / wall * 0.1, cyc / tiles, wmin / 100., wmax / 100.);
  return best;
}

template <int DIN, int DOUT, bool SIGNS, int KO = 0>
static float run16(const char* name, const float* in, const int32_t* idx, int n_sel, const float* w, float* out, uint32_t* signs,
                   int reps, int grid) {
  if (!g_stamps) CK(hipMalloc(&g_stamps, 8 * 8 * 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = (size_t)(DIN * DOUT + 64) * 4;
  auto k1 = ws16_kernel<DIN, DOUT, true, SIGNS, KO>;
  auto k0 = ws16_kernel<DIN, DOUT, false, SIGNS, KO>;
  CK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto launch = [&]() {
    if (idx) hipLaunchKernelGGL(k1, dim3(grid), dim3(256), lds, 0, in, (int64_t)DIN, idx, n_sel, w, out, (int64_t)DOUT, signs, g_stamps);
    else hipLaunchKernelGGL(k0, dim3(grid), dim3(256), lds, 0, in, (int64_t)DIN, idx, n_sel, w, out, (int64_t)DOUT, signs, g_stamps);
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0.f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms);
    sum += ms;
  }
  std::vector<unsigned long long> st(3 * grid * 4);
  CK(hipMemcpy(st.data(), g_stamps, st.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, wall = 0, tiles = 0, wmax = 0, wmin = 1e18;
  double pro = 0, fillc = 0;
  {
    std::vector<unsigned long long> st2(1024);
    CK(hipMemcpy(st2.data(), g_stamps + 3 * 1024, 1024 * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < 1024; ++i) fillc += (double)st2[i];
    fillc /= 1024;
  }
  for (int i = 0; i < grid * 4; ++i) {
    cyc += st[3 * i]; wall += st[3 * i + 1]; tiles += (double)(st[3 * i + 2] & 0xffffffffull); pro += (double)(st[3 * i + 2] >> 32);
    wmax = fmax(wmax, (double)st[3 * i + 1]); wmin = fmin(wmin, (double)st[3 * i + 1]);
  }
  const double gf = 2.0 * n_sel * DIN * DOUT * 1e-9;
  printf("%-34s rows %7d  %3d->%3d  KO %d : best %7.1f us  mean %7.1f us  = %6.1f TF | clock %.2f GHz, fill+barrier %.0f, prologue %.0f cycles, %.0f cycles/unit after it, wave life %.1f .. %.1f us\n",
         name, n_sel, DIN, DOUT, KO, best * 1e3, sum / reps * 1e3, gf / best, cyc / wall * 0.1, fillc, pro / (grid * 4), (cyc - pro) / tiles, wmin / 100., wmax / 100.);
  return best;
}

static double check(const std::vector<float>& h_in, const std::vector<float>& h_w, const std::vector<int32_t>& h_idx, int n, int n_all,
                    const float* d_out, const uint32_t* d_sg, int din, int dout, bool signs) {
  std::vector<float> h_out((size_t)n_all * dout);
  CK(hipMemcpy(h_out.data(), d_out, h_out.size() * 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> h_sg((size_t)n * (dout / 32));
  if (signs) CK(hipMemcpy(h_sg.data(), d_sg, h_sg.size() * 4, hipMemcpyDeviceToHost));
  double num = 0, den = 0;
  long bad_bits = 0;
  auto one = [&](int s) {
    const int row = h_idx[s];
    for (int j = 0; j < dout; ++j) {
      double ref = 0;
      for (int k = 0; k < din; ++k) ref += (double)h_in[(size_t)row * din + k] * h_w[k * dout + j];
      const float got = h_out[(size_t)row * dout + j];
      const double d = got - ref;
      num += d * d; den += ref * ref;
      if (signs && (((h_sg[(size_t)s * (dout / 32) + j / 32] >> (j & 31)) & 1u) != (got > 0.f ? 1u : 0u))) ++bad_bits;
    }
  };
  for (int s = 0; s < n; s += 997) one(s);
  for (int s = n - 40; s < n; ++s) one(s);
  for (int s = 0; s < 40; ++s) one(s);
  if (signs) printf("  sign bits that disagree with the stored outputs: %ld\n", bad_bits);
  return sqrt(num / den);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 178921, reps = argc > 2 ? atoi(argv[2]) : 30;
  const int n_all = 235868;                      // rows of the table the index list selects from
  std::vector<float> h_in((size_t)n_all * 128), h_w(128 * 128);
  std::vector<int32_t> h_idx(n);
  uint32_t st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : h_in) v = rnd();
  for (auto& v : h_w) v = rnd() * 0.2f;
  for (int i = 0; i < n; ++i) h_idx[i] = (int32_t)((int64_t)i * n_all / n);     // sorted subset, like the Del row lists
  float *d_in, *d_w, *d_out;
  int32_t* d_idx;
  uint32_t* d_sg;
  CK(hipMalloc(&d_in, h_in.size() * 4)); CK(hipMalloc(&d_w, h_w.size() * 4)); CK(hipMalloc(&d_out, (size_t)n_all * 128 * 4));
  CK(hipMalloc(&d_idx, (size_t)n * 4)); CK(hipMalloc(&d_sg, (size_t)n * 16));
  CK(hipMemcpy(d_in, h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_w, h_w.data(), h_w.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_idx, h_idx.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_out, 0, (size_t)n_all * 128 * 4));

  run<4, 128, 0, false>("ws 128x128 rotate4", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  CK(hipMemset(d_out, 0, (size_t)n_all * 128 * 4));
  run_ring<4, 128, 4, false>("ring 128x128 4 chunks", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  // correctness of the plain form against an fp64 product on a sample of rows
  {
    std::vector<float> h_out((size_t)n_all * 128);
    CK(hipMemcpy(h_out.data(), d_out, h_out.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (int s = 0; s < n; s += 997) {
      const int row = h_idx[s];
      for (int j = 0; j < 128; ++j) {
        double ref = 0;
        for (int k = 0; k < 128; ++k) ref += (double)h_in[(size_t)row * 128 + k] * h_w[k * 128 + j];
        const double d = h_out[(size_t)row * 128 + j] - ref;
        num += d * d; den += ref * ref;
      }
    }
    // last rows too (tail tile)
    for (int s = n - 40; s < n; ++s) {
      const int row = h_idx[s];
      for (int j = 0; j < 128; ++j) {
        double ref = 0;
        for (int k = 0; k < 128; ++k) ref += (double)h_in[(size_t)row * 128 + k] * h_w[k * 128 + j];
        const double d = h_out[(size_t)row * 128 + j] - ref;
        num += d * d; den += ref * ref;
      }
    }
    printf("  rel-L2 vs fp64 (sampled rows + tail): %.3e\n", sqrt(num / den));
  }
  CK(hipMemset(d_out, 0, (size_t)n_all * 128 * 4));
  run16<128, 128, false>("ws16 128x128", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  printf("  ws16 rel-L2 vs fp64: %.3e\n", check(h_in, h_w, h_idx, n, n_all, d_out, d_sg, 128, 128, false));
  CK(hipMemset(d_out, 0, (size_t)n_all * 128 * 4));
  run16<128, 128, true>("ws16 128x128 + signs", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  printf("  ws16 + signs rel-L2 vs fp64: %.3e\n", check(h_in, h_w, h_idx, n, n_all, d_out, d_sg, 128, 128, true));
  run16<128, 128, false, 1>("ws16 128x128 no stores", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  run16<128, 128, false, 3>("ws16 128x128 no loads no stores", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  run16<128, 128, false>("ws16 128x128 dense N rows", d_in, nullptr, n_all, d_w, d_out, d_sg, reps, 256);
  run16<128, 128, false>("ws16 128x128 196608 rows", d_in, nullptr, 196608, d_w, d_out, d_sg, reps, 256);
  run16<128, 64, false>("ws16 128x64 dense N rows", d_in, nullptr, n_all, d_w, d_out, d_sg, reps, 256);
  run16<64, 128, false>("ws16 64x128", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  run16<64, 64, false>("ws16 64x64", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  run_ring<4, 128, 4, false, 1>("ring 128x128 no stores", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  run_ring<4, 128, 4, false, 2>("ring 128x128 no loads", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  run_ring<4, 128, 4, false, 3>("ring 128x128 no loads, no stores", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  run_ring<4, 128, 4, false, 3>("ring 196608 no loads, no stores", d_in, nullptr, 196608, d_w, d_out, d_sg, reps, 256);
  run_ring<4, 128, 4, false, 1>("ring 196608 no stores", d_in, nullptr, 196608, d_w, d_out, d_sg, reps, 256);
  run_ring<4, 128, 4, false, 2>("ring 196608 no loads", d_in, nullptr, 196608, d_w, d_out, d_sg, reps, 256);
  run_ring<4, 128, 2, false>("ring 128x128 2 chunks", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  run_ring<4, 128, 8, false>("ring 128x128 8 chunks", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  run_ring<4, 128, 4, true>("ring 128x128 4 chunks + signs", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  run_ring<4, 128, 4, false>("ring 128x128 4 ch dense N rows", d_in, nullptr, n_all, d_w, d_out, d_sg, reps, 256);
  run_ring<4, 128, 4, false>("ring 128x128 4 ch 196608 rows", d_in, nullptr, 196608, d_w, d_out, d_sg, reps, 256);
  run_ring<2, 128, 4, false>("ring 128x64  4 ch dense N rows", d_in, nullptr, n_all, d_w, d_out, d_sg, reps, 256);
  run_ring<4, 64, 4, false>("ring 64x128  4 ch", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  run<4, 128, 1, false>("ws 128x128 pairs", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  run<4, 128, 0, true>("ws 128x128 rotate4 + signs", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  run<4, 128, 1, true>("ws 128x128 pairs + signs", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  run<4, 128, 0, false>("ws 128x128 rotate4 dense N rows", d_in, nullptr, n_all, d_w, d_out, d_sg, reps, 256);
  run<4, 128, 1, false>("ws 128x128 pairs   dense N rows", d_in, nullptr, n_all, d_w, d_out, d_sg, reps, 256);
  run<2, 128, 0, false>("ws 128x64  rotate2 dense N rows", d_in, nullptr, n_all, d_w, d_out, d_sg, reps, 256);
  run<4, 64, 0, false>("ws 64x128  rotate4", d_in, d_idx, n, d_w, d_out, d_sg, reps, 256);
  // exact multiples of the wave count (no tail): 6 tiles per wave
  run<4, 128, 0, false>("ws 128x128 rotate4 196608 rows", d_in, nullptr, 196608, d_w, d_out, d_sg, reps, 256);
  run<4, 128, 1, false>("ws 128x128 pairs   196608 rows", d_in, nullptr, 196608, d_w, d_out, d_sg, reps, 256);
  return 0;
}
