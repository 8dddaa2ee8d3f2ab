// Lab: the dense 128 -> 128 row GEMM (the Del operator's shape) with every fp32 product formed from bf16 partial products
// on v_mfma_f32_32x32x16_bf16 (fp32 accumulation) - "split" arithmetic: x = x1 + x2 + x3 with x_i the successive
// round-to-nearest bf16 pieces (3 x 8 significant bits + signs: exact for every fp32 value in range), so
//   NP = 9: all nine partial products s_i w_j - each exact in fp32, the sum is the fp32 product before accumulation rounding;
//   NP = 6: without s2 w3, s3 w2, s3 w3 (each <= 2^-26 |s w|);   NP = 3: s1 w1 + s1 w2 + s2 w1 (error ~2^-17);
//   NP = 1: the production form, v_mfma_f32_32x32x2_f32.
// Question asked: under the chip's power limit, is the split form faster than the fp32 matrix instruction, and how
// accurate is it against an fp64 product?   hipcc -O3 --offload-arch=gfx950 split_lab.hip -o split_lab.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512, kWaves = 8;

__device__ inline uint32_t pk(floatx2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }
__device__ inline floatx2 unpk(uint32_t p) {
  floatx2 r;
  r[0] = __builtin_bit_cast(float, p << 16);
  r[1] = __builtin_bit_cast(float, p & 0xffff0000u);
  return r;
}

// 8 floats -> NS bf16x8 pieces
template <int NS>
__device__ inline void split8(const float4 a, const float4 b, bf16x8 (&s)[NS]) {
  const floatx2 v[4] = {{a.x, a.y}, {a.z, a.w}, {b.x, b.y}, {b.z, b.w}};
  u32x4 p[NS];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    floatx2 r = v[i];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const uint32_t w = pk(r);
      p[q][i] = w;
      if (q + 1 < NS) r = r - unpk(w);
    }
  }
#pragma unroll
  for (int q = 0; q < NS; ++q) s[q] = __builtin_bit_cast(bf16x8, p[q]);
}

template <int NP, int KO = 0>      // KO (timing only, wrong results): 1 no split arithmetic, 2 no LDS reads in the loop, 3 both, 4 one MFMA per product group
__global__ __launch_bounds__(kThreads, 2) void split_gemm_kernel(const float* __restrict__ in, int n_rows,
                                                                const float* __restrict__ w, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int NS = NP == 3 ? 2 : 3;                      // pieces of each operand
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r_lo = lane & 31, khalf = lane >> 5;
  // weight image: piece p | k chunk kc (32) | m (8-k group pair) | output tile t | khalf | feature r | 8 bf16
  __bf16* const wsp = reinterpret_cast<__bf16*>(lds_raw);
  // one (8-k group, output feature) item per thread and step: 8 strided loads in flight (coalesced across the threads'
  // features), NS 16-byte LDS stores
  for (int e = tid; e < 16 * 128; e += kThreads) {
    const int kg = e >> 7, n = e & 127;                    // k = 8 kg .. 8 kg + 7
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = w[(8 * kg + c) * 128 + n];
    bf16x8 pc[NS];
    split8<NS>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), pc);
    const int off = ((((kg >> 2) * 2 + (kg & 1)) * 4 + (n >> 5)) * 2 + ((kg >> 1) & 1)) * 32 + (n & 31);
#pragma unroll
    for (int q = 0; q < NS; ++q) reinterpret_cast<bf16x8*>(wsp)[q * 2048 + off] = pc[q];
  }
  __syncthreads();
  const int n_tiles = (n_rows + 31) >> 5, stride = gridDim.x * kWaves;
  int tile = blockIdx.x * kWaves + wave;
  if (tile >= n_tiles) return;
  auto row_of = [&](int t) { return min(t * 32 + r_lo, n_rows - 1); };
  // whole half rows in registers, the NEXT tile's in flight while this one feeds the matrix cores (a k chunk of the split
  // form lasts < 1 us: chunk-wise prefetch no longer covers the HBM latency)
  const bf16x8* const wv = reinterpret_cast<const bf16x8*>(wsp);
  auto fetch = [&](int t, float4 (&a)[16]) {
    const float4* s0 = reinterpret_cast<const float4*>(in + (int64_t)row_of(min(t, n_tiles - 1)) * 128) + khalf * 4;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
      for (int i = 0; i < 4; ++i) a[kc * 4 + i] = s0[kc * 8 + i];
  };
  auto work = [&](int t_, const float4 (&a)[16]) {
    const int row = t_ * 32 + r_lo;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        bf16x8 s[NS];
        if (KO & 1) {
#pragma unroll
          for (int q = 0; q < NS; ++q) s[q] = __builtin_bit_cast(bf16x8, a[kc * 4 + 2 * m + (q & 1)]);
        } else {
          split8<NS>(a[kc * 4 + 2 * m], a[kc * 4 + 2 * m + 1], s);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int o = (((kc * 2 + m) * 4 + t) * 2 + khalf) * 32 + r_lo;
          bf16x8 wq[NS];
#pragma unroll
          for (int q = 0; q < NS; ++q) wq[q] = (KO & 2) ? __builtin_bit_cast(bf16x8, a[(q + t) & 15]) : wv[q * 2048 + o];
          if (NP == 9) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[NS - 1], s[NS - 1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[NS - 1], s[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[1], s[NS - 1], acc[t], 0, 0, 0);
          }
          if (NP >= 6) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[NS - 1], s[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[1], s[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[0], s[NS - 1], acc[t], 0, 0, 0);
          }
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[1], s[0], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[0], s[1], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[0], s[0], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                 // keep the LDS reads of later k groups where they are
      }
    }
    if (row < n_rows) {
      float* dst = out + (int64_t)row * 128 + 4 * khalf;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(dst + 32 * t + 8 * q) = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
    }
  };
  float4 ra[16], rb[16];
  fetch(tile, ra);
  for (; tile < n_tiles; tile += 2 * stride) {
    fetch(tile + stride, rb);
    work(tile, ra);
    if (tile + stride >= n_tiles) break;
    fetch(tile + 2 * stride, ra);
    work(tile + stride, rb);
  }
}

// the production form: fp32 matrix instruction, tile-interleaved fp32 weight image
__device__ unsigned long long g_clk[4];
__device__ unsigned long long g_ph[16];
__device__ unsigned long long g_wave[4 * 8192];      // per wave: hw id | xcc id << 32, block * 8 + wave, wall-clock start, end
__device__ unsigned long long g_span[2 * 1024];     // wall-clock start / end of wave 0 of every block       // phase stamps of block 7 wave 0: start, after fill, then (k loop end, stores issued) per tile
__global__ __launch_bounds__(kThreads, 4) void f32_gemm_kernel(const float* __restrict__ in, int n_rows, const float* __restrict__ w,
                                                              float* __restrict__ out) {
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const wl = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r_lo = lane & 31, khalf = lane >> 5;
  for (int e = tid; e < 128 * 32; e += kThreads) {
    const int k = e >> 5, r = e & 31;
    *reinterpret_cast<float4*>(wl + e * 4) = make_float4(w[k * 128 + r], w[k * 128 + 32 + r], w[k * 128 + 64 + r], w[k * 128 + 96 + r]);
  }
  __syncthreads();
  const bool probe = blockIdx.x == 7 && threadIdx.x == 0;
  int ph = 0;
  if (probe) { g_ph[ph++] = c0; g_ph[ph++] = __builtin_readcyclecounter(); }
  const int n_tiles = (n_rows + 31) >> 5, stride = gridDim.x * kWaves;
  int tile = blockIdx.x * kWaves + wave;
  if (tile >= n_tiles) return;
  auto row_of = [&](int t) { return min(t * 32 + r_lo, n_rows - 1); };
  int row_cur = row_of(tile), row_nxt = row_of(min(tile + stride, n_tiles - 1));
  float4 a_next[4];
  {
    const float4* s0 = reinterpret_cast<const float4*>(in + (int64_t)row_cur * 128) + khalf * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) a_next[i] = s0[i];
  }
  for (; tile < n_tiles; tile += stride) {
    const bool live = tile * 32 + r_lo < n_rows;
    const float4* src = reinterpret_cast<const float4*>(in + (int64_t)row_cur * 128) + khalf * 4;
    const float4* src_n = reinterpret_cast<const float4*>(in + (int64_t)row_nxt * 128) + khalf * 4;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll 1
    for (int kc = 0; kc < 4; ++kc) {
      float4 a4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a4[i] = a_next[i];
      const float4* nsrc = kc + 1 < 4 ? src + (kc + 1) * 8 : src_n;
#pragma unroll
      for (int i = 0; i < 4; ++i) a_next[i] = nsrc[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float av[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float4 f = *reinterpret_cast<const float4*>(wl + ((kc * 32 + khalf * 16 + i * 4 + s) * 32 + r_lo) * 4);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, av[s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, av[s], acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, av[s], acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, av[s], acc[3], 0, 0, 0);
        }
      }
    }
    if (probe && ph < 14) g_ph[ph++] = __builtin_readcyclecounter();
    float* dst = out + (int64_t)row_cur * 128 + 4 * khalf;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (live) *reinterpret_cast<float4*>(dst + 32 * t + 8 * q) = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
    if (probe && ph < 14) g_ph[ph++] = __builtin_readcyclecounter();
    row_cur = row_nxt;
    row_nxt = row_of(min(tile + 2 * stride, n_tiles - 1));
  }
  if (probe) g_ph[15] = ph;
  if (lane == 0 && blockIdx.x * 8 + wave < 8192) {
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* rec = g_wave + 4 * (blockIdx.x * 8 + wave);
    rec[0] = hw | ((unsigned long long)(xcc & 15) << 32);
    rec[1] = blockIdx.x * 8 + wave;
    rec[2] = w0;
    rec[3] = wall_clock64();
  }
  if (threadIdx.x == 0 && blockIdx.x < 1024) { g_span[2 * blockIdx.x] = w0; g_span[2 * blockIdx.x + 1] = wall_clock64(); }
  if (blockIdx.x == 7 && threadIdx.x == 0) { g_clk[0] = __builtin_readcyclecounter() - c0; g_clk[1] = wall_clock64() - w0; }
}

// the production form: fp32 matrix instruction, tile-interleaved fp32 weight image
// NON-transposed product (sample rows = D rows): a lane ends with ONE feature of 16 samples, and every store instruction
// writes 128 contiguous bytes of two rows (dword per lane) instead of 32-byte pieces of 32 rows
__global__ __launch_bounds__(kThreads, 4) void f32_rowmajor_kernel(const float* __restrict__ in, int n_rows, const float* __restrict__ w,
                                                              float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const wl = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r_lo = lane & 31, khalf = lane >> 5;
  for (int e = tid; e < 128 * 32; e += kThreads) {
    const int k = e >> 5, r = e & 31;
    *reinterpret_cast<float4*>(wl + e * 4) = make_float4(w[k * 128 + r], w[k * 128 + 32 + r], w[k * 128 + 64 + r], w[k * 128 + 96 + r]);
  }
  __syncthreads();
  const int n_tiles = (n_rows + 31) >> 5, stride = gridDim.x * kWaves;
  int tile = blockIdx.x * kWaves + wave;
  if (tile >= n_tiles) return;
  auto row_of = [&](int t) { return min(t * 32 + r_lo, n_rows - 1); };
  int row_cur = row_of(tile), row_nxt = row_of(min(tile + stride, n_tiles - 1));
  float4 a_next[4];
  {
    const float4* s0 = reinterpret_cast<const float4*>(in + (int64_t)row_cur * 128) + khalf * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) a_next[i] = s0[i];
  }
  for (; tile < n_tiles; tile += stride) {
    const bool live = tile * 32 + r_lo < n_rows;
    const float4* src = reinterpret_cast<const float4*>(in + (int64_t)row_cur * 128) + khalf * 4;
    const float4* src_n = reinterpret_cast<const float4*>(in + (int64_t)row_nxt * 128) + khalf * 4;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll 1
    for (int kc = 0; kc < 4; ++kc) {
      float4 a4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a4[i] = a_next[i];
      const float4* nsrc = kc + 1 < 4 ? src + (kc + 1) * 8 : src_n;
#pragma unroll
      for (int i = 0; i < 4; ++i) a_next[i] = nsrc[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float av[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float4 f = *reinterpret_cast<const float4*>(wl + ((kc * 32 + khalf * 16 + i * 4 + s) * 32 + r_lo) * 4);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], f.x, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], f.y, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], f.z, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], f.w, acc[3], 0, 0, 0);
        }
      }
    }
    // D[i][j]: i = sample (r & 3) + 8 (r >> 2) + 4 khalf, j = lane & 31 = feature within tile t
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * khalf;
      const int rowi = __shfl(row_cur, i);
      if (tile * 32 + i < n_rows) {
        float* dst = out + (int64_t)rowi * 128 + r_lo;
#pragma unroll
        for (int t = 0; t < 4; ++t) dst[32 * t] = acc[t][r];
      }
    }
    row_cur = row_nxt;
    row_nxt = row_of(min(tile + 2 * stride, n_tiles - 1));
  }
}

// the production form: fp32 matrix instruction, tile-interleaved fp32 weight image
// one block of 16 waves per CU sharing ONE weight image, the block's tiles handed out through an LDS counter: a wave that
// the (oldest-first) arbitration of the matrix pipe lets run ahead simply takes more tiles
__global__ __launch_bounds__(1024, 4) void f32_queue_kernel(const float* __restrict__ in, int n_rows, const float* __restrict__ w,
                                                              float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const wl = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r_lo = lane & 31, khalf = lane >> 5;
  for (int e = tid; e < 128 * 32; e += 1024) {
    const int k = e >> 5, r = e & 31;
    *reinterpret_cast<float4*>(wl + e * 4) = make_float4(w[k * 128 + r], w[k * 128 + 32 + r], w[k * 128 + 64 + r], w[k * 128 + 96 + r]);
  }
  __syncthreads();
  __shared__ int q_next;
  const int n_tiles = (n_rows + 31) >> 5;
  const int per_block = (n_tiles + gridDim.x - 1) / gridDim.x, t_lo = blockIdx.x * per_block, t_hi = min(n_tiles, t_lo + per_block);
  if (tid == 0) q_next = 0;
  __syncthreads();
  auto grab = [&]() -> int {
    int t = 0;
    if (lane == 0) t = atomicAdd(&q_next, 1);
    t = __builtin_amdgcn_readfirstlane(t) + t_lo;
    return t < t_hi ? t : n_tiles;
  };
  int tile = grab();
  if (tile >= n_tiles) return;
  int tile_nxt = grab();
  auto row_of = [&](int t) { return min(t * 32 + r_lo, n_rows - 1); };
  int row_cur = row_of(tile), row_nxt = row_of(min(tile_nxt, n_tiles - 1));
  float4 a_next[4];
  {
    const float4* s0 = reinterpret_cast<const float4*>(in + (int64_t)row_cur * 128) + khalf * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) a_next[i] = s0[i];
  }
  for (; tile < n_tiles;) {
    const bool live = tile * 32 + r_lo < n_rows;
    const float4* src = reinterpret_cast<const float4*>(in + (int64_t)row_cur * 128) + khalf * 4;
    const float4* src_n = reinterpret_cast<const float4*>(in + (int64_t)row_nxt * 128) + khalf * 4;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll 1
    for (int kc = 0; kc < 4; ++kc) {
      float4 a4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a4[i] = a_next[i];
      const float4* nsrc = kc + 1 < 4 ? src + (kc + 1) * 8 : src_n;
#pragma unroll
      for (int i = 0; i < 4; ++i) a_next[i] = nsrc[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float av[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float4 f = *reinterpret_cast<const float4*>(wl + ((kc * 32 + khalf * 16 + i * 4 + s) * 32 + r_lo) * 4);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, av[s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, av[s], acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, av[s], acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, av[s], acc[3], 0, 0, 0);
        }
      }
    }
    float* dst = out + (int64_t)row_cur * 128 + 4 * khalf;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (live) *reinterpret_cast<float4*>(dst + 32 * t + 8 * q) = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
    tile = tile_nxt;
    tile_nxt = tile < n_tiles ? grab() : n_tiles;
    row_cur = row_nxt;
    row_nxt = row_of(min(tile_nxt, n_tiles - 1));
  }
}

// the production form: fp32 matrix instruction, tile-interleaved fp32 weight image
// STAG: the waves that share a SIMD start STAG x 127 x 64 cycles apart, so that their tile epilogues (16 row stores each) stop coinciding
template <int STAG>
__global__ __launch_bounds__(kThreads, 4) void f32_stag_kernel_(const float* __restrict__ in, int n_rows, const float* __restrict__ w,
                                                              float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const wl = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r_lo = lane & 31, khalf = lane >> 5;
  if (STAG == 3) {
    // fast fill: 8 coalesced 16-byte loads per thread, all in flight at once, scattered into the interleaved image
    float4 v[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) v[it] = reinterpret_cast<const float4*>(w)[tid + it * kThreads];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int e4 = tid + it * kThreads, k = e4 >> 5, n0 = (e4 & 31) * 4, t = n0 >> 5, r = n0 & 31;
      float* dst = wl + (k * 32 + r) * 4 + t;
      dst[0] = v[it].x; dst[4] = v[it].y; dst[8] = v[it].z; dst[12] = v[it].w;
    }
  } else
  for (int e = tid; e < 128 * 32; e += kThreads) {
    const int k = e >> 5, r = e & 31;
    *reinterpret_cast<float4*>(wl + e * 4) = make_float4(w[k * 128 + r], w[k * 128 + 32 + r], w[k * 128 + 64 + r], w[k * 128 + 96 + r]);
  }
  __syncthreads();
  {
    const int phase = STAG == 3 ? 0 : (wave >> 2) + 2 * ((blockIdx.x >> 8) & 1);
    for (int i = 0; i < phase * STAG; ++i) __builtin_amdgcn_s_sleep(127);
  }
  const int n_tiles = (n_rows + 31) >> 5, stride = gridDim.x * kWaves;
  int tile = blockIdx.x * kWaves + wave;
  if (tile >= n_tiles) return;
  auto row_of = [&](int t) { return min(t * 32 + r_lo, n_rows - 1); };
  int row_cur = row_of(tile), row_nxt = row_of(min(tile + stride, n_tiles - 1));
  float4 a_next[4];
  {
    const float4* s0 = reinterpret_cast<const float4*>(in + (int64_t)row_cur * 128) + khalf * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) a_next[i] = s0[i];
  }
  for (; tile < n_tiles; tile += stride) {
    const bool live = tile * 32 + r_lo < n_rows;
    const float4* src = reinterpret_cast<const float4*>(in + (int64_t)row_cur * 128) + khalf * 4;
    const float4* src_n = reinterpret_cast<const float4*>(in + (int64_t)row_nxt * 128) + khalf * 4;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll 1
    for (int kc = 0; kc < 4; ++kc) {
      float4 a4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a4[i] = a_next[i];
      const float4* nsrc = kc + 1 < 4 ? src + (kc + 1) * 8 : src_n;
#pragma unroll
      for (int i = 0; i < 4; ++i) a_next[i] = nsrc[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float av[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float4 f = *reinterpret_cast<const float4*>(wl + ((kc * 32 + khalf * 16 + i * 4 + s) * 32 + r_lo) * 4);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, av[s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, av[s], acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, av[s], acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, av[s], acc[3], 0, 0, 0);
        }
      }
    }
    float* dst = out + (int64_t)row_cur * 128 + 4 * khalf;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (live) *reinterpret_cast<float4*>(dst + 32 * t + 8 * q) = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
    row_cur = row_nxt;
    row_nxt = row_of(min(tile + 2 * stride, n_tiles - 1));
  }
}

// the production form: fp32 matrix instruction, tile-interleaved fp32 weight image
__global__ __launch_bounds__(kThreads, 4) void f32_pf_kernel(const float* __restrict__ in, int n_rows, const float* __restrict__ w,
                                                              float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const wl = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r_lo = lane & 31, khalf = lane >> 5;
  for (int e = tid; e < 128 * 32; e += kThreads) {
    const int k = e >> 5, r = e & 31;
    *reinterpret_cast<float4*>(wl + e * 4) = make_float4(w[k * 128 + r], w[k * 128 + 32 + r], w[k * 128 + 64 + r], w[k * 128 + 96 + r]);
  }
  __syncthreads();
  const int n_tiles = (n_rows + 31) >> 5, stride = gridDim.x * kWaves;
  int tile = blockIdx.x * kWaves + wave;
  if (tile >= n_tiles) return;
  auto row_of = [&](int t) { return min(t * 32 + r_lo, n_rows - 1); };
  int row_cur = row_of(tile), row_nxt = row_of(min(tile + stride, n_tiles - 1));
  float4 a_next[4];
  {
    const float4* s0 = reinterpret_cast<const float4*>(in + (int64_t)row_cur * 128) + khalf * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) a_next[i] = s0[i];
  }
  for (; tile < n_tiles; tile += stride) {
    const bool live = tile * 32 + r_lo < n_rows;
    const float4* src = reinterpret_cast<const float4*>(in + (int64_t)row_cur * 128) + khalf * 4;
    const float4* src_n = reinterpret_cast<const float4*>(in + (int64_t)row_nxt * 128) + khalf * 4;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll 1
    for (int kc = 0; kc < 4; ++kc) {
      float4 a4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a4[i] = a_next[i];
      const float4* nsrc = kc + 1 < 4 ? src + (kc + 1) * 8 : src_n;
#pragma unroll
      for (int i = 0; i < 4; ++i) a_next[i] = nsrc[i];
      // weight operands one k step AHEAD of the matrix instructions that use them (the wave never waits for the LDS)
      const float* wbase = wl + ((kc * 32 + khalf * 16) * 32 + r_lo) * 4;
      float4 fq[2];
      fq[0] = *reinterpret_cast<const float4*>(wbase);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (e + 1 < 16) fq[(e + 1) & 1] = *reinterpret_cast<const float4*>(wbase + (e + 1) * 128);
        const float4 f = fq[e & 1];
        const float4 a = a4[e >> 2];
        const float av = (e & 3) == 0 ? a.x : (e & 3) == 1 ? a.y : (e & 3) == 2 ? a.z : a.w;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, av, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, av, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, av, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, av, acc[3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float* dst = out + (int64_t)row_cur * 128 + 4 * khalf;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (live) *reinterpret_cast<float4*>(dst + 32 * t + 8 * q) = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
    row_cur = row_nxt;
    row_nxt = row_of(min(tile + 2 * stride, n_tiles - 1));
  }
}

template <int KO>   // 1: operand rows not loaded (registers), 2: weight operands not read from LDS, 4: no stores
__global__ __launch_bounds__(kThreads, 4) void f32_ko_kernel(const float* __restrict__ in, int n_rows, const float* __restrict__ w,
                                                              float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const wl = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r_lo = lane & 31, khalf = lane >> 5;
  for (int e = tid; e < 128 * 32; e += kThreads) {
    const int k = e >> 5, r = e & 31;
    *reinterpret_cast<float4*>(wl + e * 4) = make_float4(w[k * 128 + r], w[k * 128 + 32 + r], w[k * 128 + 64 + r], w[k * 128 + 96 + r]);
  }
  __syncthreads();
  const int n_tiles = (n_rows + 31) >> 5, stride = gridDim.x * kWaves;
  int tile = blockIdx.x * kWaves + wave;
  if (tile >= n_tiles) return;
  auto row_of = [&](int t) { return min(t * 32 + r_lo, n_rows - 1); };
  int row_cur = row_of(tile), row_nxt = row_of(min(tile + stride, n_tiles - 1));
  float4 a_next[4];
  {
    const float4* s0 = reinterpret_cast<const float4*>(in + (int64_t)row_cur * 128) + khalf * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) a_next[i] = s0[i];
  }
  for (; tile < n_tiles; tile += stride) {
    const bool live = tile * 32 + r_lo < n_rows;
    const float4* src = reinterpret_cast<const float4*>(in + (int64_t)row_cur * 128) + khalf * 4;
    const float4* src_n = reinterpret_cast<const float4*>(in + (int64_t)row_nxt * 128) + khalf * 4;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll 1
    for (int kc = 0; kc < 4; ++kc) {
      float4 a4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a4[i] = a_next[i];
      const float4* nsrc = kc + 1 < 4 ? src + (kc + 1) * 8 : src_n;
#pragma unroll
      for (int i = 0; i < 4; ++i) if (!(KO & 1)) a_next[i] = nsrc[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float av[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float4 f = (KO & 2) ? a4[(i + s) & 3] : *reinterpret_cast<const float4*>(wl + ((kc * 32 + khalf * 16 + i * 4 + s) * 32 + r_lo) * 4);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, av[s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, av[s], acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, av[s], acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, av[s], acc[3], 0, 0, 0);
        }
      }
    }
    float* dst = out + (int64_t)row_cur * 128 + 4 * khalf;
    if (KO & 4) {                                           // ONE store per tile instead of 16 (everything folded into it)
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) { sum.x += acc[t][4 * q]; sum.y += acc[t][4 * q + 1]; sum.z += acc[t][4 * q + 2]; sum.w += acc[t][4 * q + 3]; }
      if (live) *reinterpret_cast<float4*>(dst) = sum;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (live && !(KO & 4)) *reinterpret_cast<float4*>(dst + 32 * t + 8 * q) = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
    row_cur = row_nxt;
    row_nxt = row_of(min(tile + 2 * stride, n_tiles - 1));
  }
}

static float frand(uint64_t& s) {                           // roughly normal, a few orders of magnitude of spread
  float a = 0.f;
  for (int i = 0; i < 4; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; a += (float)((s >> 33) & 0xffffff) / 16777216.f - 0.5f; }
  return a;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 178921, reps = argc > 2 ? atoi(argv[2]) : 30;
  std::vector<float> hx((size_t)n * 128), hw(128 * 128), ho((size_t)n * 128);
  uint64_t seed = 42;
  for (auto& v : hx) v = frand(seed) * (1.f + 7.f * (frand(seed) > 0.4f));
  for (auto& v : hw) v = 0.1f * frand(seed);
  float *x, *w, *o;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&o, ho.size() * 4));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  const int check = 512;
  std::vector<double> ref((size_t)check * 128);
  for (int r = 0; r < check; ++r)
    for (int c = 0; c < 128; ++c) {
      double a = 0;
      for (int k = 0; k < 128; ++k) a += (double)hx[(size_t)r * 128 + k] * (double)hw[k * 128 + c];
      ref[(size_t)r * 128 + c] = a;
    }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) {
    CK(hipMemset(o, 0, ho.size() * 4));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(ho.data(), o, (size_t)check * 128 * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0, worst = 0;
    for (size_t i = 0; i < ref.size(); ++i) {
      const double d = ho[i] - ref[i];
      num += d * d; den += ref[i] * ref[i];
      worst = fmax(worst, fabs(d));
    }
    const double us = 1e3 * ms / reps, flops = 2.0 * n * 128 * 128;
    printf("%-34s %7.1f us  %6.1f TF fp32-equivalent  %5.0f GB/s   rel-L2 vs fp64 %.2e  max abs %.2e\n", name, us, flops / us / 1e6,
           2.0 * n * 512 / us / 1e3, sqrt(num / den), worst);
  };
  const int grid = argc > 3 ? atoi(argv[3]) : 256;
  CK(hipFuncSetAttribute((const void*)split_gemm_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
  CK(hipFuncSetAttribute((const void*)split_gemm_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
  CK(hipFuncSetAttribute((const void*)split_gemm_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute((const void*)f32_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute((const void*)split_gemm_kernel<6, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
  CK(hipFuncSetAttribute((const void*)split_gemm_kernel<6, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
  CK(hipFuncSetAttribute((const void*)split_gemm_kernel<6, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
  run("x6, knock-out: no split arithmetic", [&] { hipLaunchKernelGGL((split_gemm_kernel<6, 1>), dim3(grid), dim3(kThreads), 98304, 0, x, n, w, o); });
  run("x6, knock-out: no LDS reads", [&] { hipLaunchKernelGGL((split_gemm_kernel<6, 2>), dim3(grid), dim3(kThreads), 98304, 0, x, n, w, o); });
  run("x6, knock-out: neither", [&] { hipLaunchKernelGGL((split_gemm_kernel<6, 3>), dim3(grid), dim3(kThreads), 98304, 0, x, n, w, o); });
  CK(hipFuncSetAttribute((const void*)f32_ko_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute((const void*)f32_ko_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute((const void*)f32_ko_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute((const void*)f32_ko_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute((const void*)f32_ko_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  run("fp32, knock-out: operand rows in registers", [&] { hipLaunchKernelGGL(f32_ko_kernel<1>, dim3(2 * grid), dim3(kThreads), 65536, 0, x, n, w, o); });
  run("fp32, knock-out: no LDS weight reads", [&] { hipLaunchKernelGGL(f32_ko_kernel<2>, dim3(2 * grid), dim3(kThreads), 65536, 0, x, n, w, o); });
  run("fp32, knock-out: neither", [&] { hipLaunchKernelGGL(f32_ko_kernel<3>, dim3(2 * grid), dim3(kThreads), 65536, 0, x, n, w, o); });
  run("fp32, knock-out: one store per tile instead of 16", [&] { hipLaunchKernelGGL(f32_ko_kernel<4>, dim3(2 * grid), dim3(kThreads), 65536, 0, x, n, w, o); });
  run("fp32, knock-out: no loads, no LDS reads, one store per tile", [&] { hipLaunchKernelGGL(f32_ko_kernel<7>, dim3(2 * grid), dim3(kThreads), 65536, 0, x, n, w, o); });
  CK(hipFuncSetAttribute((const void*)f32_pf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  run("fp32, LDS operands one k step ahead", [&] { hipLaunchKernelGGL(f32_pf_kernel, dim3(2 * grid), dim3(kThreads), 65536, 0, x, n, w, o); });
  CK(hipFuncSetAttribute((const void*)f32_stag_kernel_<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute((const void*)f32_stag_kernel_<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  run("fp32, waves of a SIMD staggered by 8 k cycles", [&] { hipLaunchKernelGGL(f32_stag_kernel_<1>, dim3(2 * grid), dim3(kThreads), 65536, 0, x, n, w, o); });
  CK(hipFuncSetAttribute((const void*)f32_stag_kernel_<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  run("fp32, weight image filled by 8 vector loads in flight", [&] { hipLaunchKernelGGL(f32_stag_kernel_<3>, dim3(2 * grid), dim3(kThreads), 65536, 0, x, n, w, o); });
  run("fp32, waves of a SIMD staggered by 16 k cycles", [&] { hipLaunchKernelGGL(f32_stag_kernel_<2>, dim3(2 * grid), dim3(kThreads), 65536, 0, x, n, w, o); });
  CK(hipFuncSetAttribute((const void*)f32_queue_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  run("fp32, 16 waves per CU sharing an LDS tile queue", [&] { hipLaunchKernelGGL(f32_queue_kernel, dim3(grid), dim3(1024), 65536, 0, x, n, w, o); });
  CK(hipFuncSetAttribute((const void*)f32_rowmajor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  run("fp32, non-transposed product: 128-byte row segments per store", [&] { hipLaunchKernelGGL(f32_rowmajor_kernel, dim3(2 * grid), dim3(kThreads), 65536, 0, x, n, w, o); });
  for (int round = 0; round < 2; ++round) {
    run("fp32 mfma 32x32x2 (production form)", [&] { hipLaunchKernelGGL(f32_gemm_kernel, dim3(2 * grid), dim3(kThreads), 65536, 0, x, n, w, o); });
    run("bf16 split x9", [&] { hipLaunchKernelGGL(split_gemm_kernel<9>, dim3(grid), dim3(kThreads), 98304, 0, x, n, w, o); });
    run("bf16 split x6", [&] { hipLaunchKernelGGL(split_gemm_kernel<6>, dim3(grid), dim3(kThreads), 98304, 0, x, n, w, o); });
    run("bf16 split x3", [&] { hipLaunchKernelGGL(split_gemm_kernel<3>, dim3(2 * grid), dim3(kThreads), 65536, 0, x, n, w, o); });
  }
  unsigned long long clk[4];
  CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk)));
  unsigned long long ph[16];
  CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_ph), sizeof(ph)));
  printf("phases of block 7 wave 0 (shader cycles): fill %llu", ph[1] - ph[0]);
  for (int i = 2; i + 1 < (int)ph[15]; i += 2) printf(" | tile k-loop %llu epilogue %llu", ph[i] - ph[i - 1], ph[i + 1] - ph[i]);
  printf("\n");
  {
    static unsigned long long span[2048];
    CK(hipMemcpyFromSymbol(span, HIP_SYMBOL(g_span), sizeof(span)));
    unsigned long long s_min = ~0ull, s_max = 0, e_min = ~0ull, e_max = 0;
    const int nb = 2 * grid;
    for (int i = 0; i < nb && i < 1024; ++i) {
      if (span[2 * i] < s_min) s_min = span[2 * i];
      if (span[2 * i] > s_max) s_max = span[2 * i];
      if (span[2 * i + 1] < e_min) e_min = span[2 * i + 1];
      if (span[2 * i + 1] > e_max) e_max = span[2 * i + 1];
    }
    printf("last fp32 launch, wave 0 of %d blocks (us, from the first start): last start %.1f, first end %.1f, last end %.1f\n", nb,
           (s_max - s_min) * 0.01, (e_min - s_min) * 0.01, (e_max - s_min) * 0.01);
    printf("  mean / min / max end per XCD (block index mod 8):");
    for (int x = 0; x < 8; ++x) {
      double sum = 0, mn = 1e30, mx = 0;
      int cnt = 0;
      for (int i = x; i < nb && i < 1024; i += 8) { const double e = (span[2 * i + 1] - s_min) * 0.01; sum += e; mn = e < mn ? e : mn; mx = e > mx ? e : mx; ++cnt; }
      printf("  [%d] %.0f / %.0f / %.0f", x, sum / cnt, mn, mx);
    }
    {
      static unsigned long long wv[4 * 8192];
      CK(hipMemcpyFromSymbol(wv, HIP_SYMBOL(g_wave), sizeof(wv)));
      const int nw = nb * 8 < 8192 ? nb * 8 : 8192;
      // waves that shared SIMD (xcc 0, se 0, sh 0, cu of block 0, simd of block 0 wave 0) and two more SIMDs
      for (int pick = 0; pick < 3; ++pick) {
        const unsigned long long key = wv[4 * (pick * 8 * 37)] & 0xf0000fff0ull;      // xcc | se sh cu | simd
        printf("\n  SIMD of wave %d (xcc %llu, hw id bits %03llx): [block.wave start-end us]", pick * 8 * 37, wv[4 * (pick * 8 * 37)] >> 32,
               (wv[4 * (pick * 8 * 37)] >> 4) & 0xfff);
        for (int i = 0; i < nw; ++i)
          if ((wv[4 * i] & 0xf0000fff0ull) == key)
            printf("  %llu.%llu %.0f-%.0f", wv[4 * i + 1] / 8, wv[4 * i + 1] % 8, (wv[4 * i + 2] - s_min) * 0.01, (wv[4 * i + 3] - s_min) * 0.01);
      }
    }
    printf("\n  ends of blocks 0..31:");
    for (int i = 0; i < 32; ++i) printf(" %.0f", (span[2 * i + 1] - s_min) * 0.01);
    printf("\n");
  }
  printf("fp32 kernel, block 7 wave 0: %llu shader cycles in %llu ticks of the 100 MHz wall clock -> %.0f MHz shader clock under this kernel\n", clk[0], clk[1],
         100.0 * (double)clk[0] / (double)clk[1]);
  return 0;
}
