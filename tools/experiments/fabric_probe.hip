// What bounds a gather of 512-byte rows on MI355X: calibration kernels for the SpMM.
//   stream   : wide coalesced reads of a buffer that lives in HBM / in the Infinity Cache / in L2
//   gather   : TL "local" trips (rows inside a +-64-row window: L1/L2 hits) + TR "remote" trips (uniformly
//              random rows of a 121 MB table: L2 misses) per output row, two rows per trip (half-wave per row),
//              optionally on a subset of the XCDs (is the miss rate a per-XCD or a chip-wide limit?)
//   ldswin   : the same row mix with the block's own row window staged in LDS (plain loads + ds_write_b128 or
//              LDS-DMA) and the local trips served by ds_read_b128
// Indices are hashed from the row id (they cost no memory traffic).  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
  return a;
}

__global__ void fill_kernel(float* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (float)(mix((uint32_t)i) & 0xffff) * (1.0f / 65536.0f) - 0.5f;
}

// ---------------------------------------------------------------- stream
template <int U>
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ buf, size_t n4, int shared,
                                                     float* __restrict__ out) {
  const int xcd = blockIdx.x & 7;
  const size_t waves_per_xcd = (size_t)(gridDim.x / 8) * 4;
  const size_t wx = (size_t)(blockIdx.x / 8) * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const size_t per = shared ? n4 : n4 / 8;
  const float4* base = buf + (shared ? 0 : (size_t)xcd * per);
  // in shared mode every XCD starts somewhere else so that they do not walk the buffer in lock step
  const size_t rot = shared ? (size_t)xcd * (per / 8) : 0;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = wx * 64 * U; i + 64 * U <= per; i += waves_per_xcd * 64 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      size_t j = i + rot + (size_t)u * 64 + lane;
      if (j >= per) j -= per;
      v[u] = base[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// ---------------------------------------------------------------- gather (L1/L2 path only)
template <int U>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int TL,
                                                     int TR, int xcd_mask, int n_active) {
  const int lane = threadIdx.x & 63, g = lane >> 5, li = lane & 31;
  const int xcd = blockIdx.x & 7;
  if (!((xcd_mask >> xcd) & 1)) return;
  const int slot = __popc(xcd_mask & ((1 << xcd) - 1));
  const int waves_per_xcd = (gridDim.x / 8) * 4;
  const int wx = (blockIdx.x / 8) * 4 + (threadIdx.x >> 6);
  const int per = (n + n_active - 1) / n_active;
  const int r0 = slot * per, r1 = min(n, r0 + per);
  const char* xb = reinterpret_cast<const char*>(x);
  const int T = TL + TR;
  for (int r = r0 + wx; r < r1; r += waves_per_xcd) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = 0; t0 < T; t0 += U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u;
        const uint32_t h = mix((uint32_t)r * 64u + (uint32_t)(2 * t + g));
        int c;
        if (t < TR) c = (int)(h % (uint32_t)n);                 // remote: anywhere in the table
        else { c = r + (int)(h % 129u) - 64; c = c < 0 ? c + n : (c >= n ? c - n : c); }
        v[u] = t < T ? *reinterpret_cast<const float4*>(xb + ((uint32_t)c * 512u + (uint32_t)li * 16u))
                     : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32);
    acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
    if (g == 0) *reinterpret_cast<float4*>(reinterpret_cast<char*>(y) + ((size_t)r * 512u + li * 16u)) = acc;
  }
}

// ---------------------------------------------------------------- gather with the row window in LDS
__device__ __forceinline__ int xcd_contiguous_block(int b, int n_blocks) {
  const int q = n_blocks / 8, r = n_blocks % 8;
  const int xcd = b % 8, local = b / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

template <int WR, int NT, bool DMA>
__global__ __launch_bounds__(NT) void ldswin_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int TL,
                                                    int TR) {
  extern __shared__ float4 win[];                           // WR rows x 32 float4
  constexpr int NW = NT / 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 5, li = lane & 31;
  const int wi = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int w0 = wi * WR;
  const int rows = min(WR, n - w0);
  const char* xb = reinterpret_cast<const char*>(x);
  // stage: a wave moves two rows (1 KB) per load
  constexpr int PER = WR / (2 * NW);
  if (DMA) {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int r2 = (p * NW + wv) * 2;                      // wave-uniform first row of the pair
      const int r = min(r2 + g, rows - 1);
      __builtin_amdgcn_global_load_lds(
          (const void*)(xb + ((uint32_t)(w0 + r) * 512u + (uint32_t)li * 16u)),
          (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(win) + r2 * 512), 16, 0, 0);
    }
  } else {
    float4 v[PER];
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int r = min((p * NW + wv) * 2 + g, rows - 1);
      v[p] = *reinterpret_cast<const float4*>(xb + ((uint32_t)(w0 + r) * 512u + (uint32_t)li * 16u));
    }
#pragma unroll
    for (int p = 0; p < PER; ++p) win[((p * NW + wv) * 2 + g) * 32 + li] = v[p];
  }
  __syncthreads();
  for (int rr = wv; rr < rows; rr += NW) {
    const int r = w0 + rr;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // remote trips first (asynchronous), local trips from LDS underneath them
    float4 rv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t h = mix((uint32_t)r * 64u + (uint32_t)(2 * t + g));
      const int c = (int)(h % (uint32_t)n);
      rv[t] = t < TR ? *reinterpret_cast<const float4*>(xb + ((uint32_t)c * 512u + (uint32_t)li * 16u))
                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int t = 0; t < TL; ++t) {
      const uint32_t h = mix((uint32_t)r * 64u + (uint32_t)(2 * (t + 4) + g));
      const int c = (int)(h % (uint32_t)rows);
      const float4 v = win[c * 32 + li];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) { acc.x += rv[t].x; acc.y += rv[t].y; acc.z += rv[t].z; acc.w += rv[t].w; }
    acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32);
    acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
    if (g == 0) *reinterpret_cast<float4*>(reinterpret_cast<char*>(y) + ((size_t)r * 512u + li * 16u)) = acc;
  }
}

template <typename F>
static double time_us(F launch, int reps = 20) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / reps * 1e3;
}

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "all";
  const bool all = !strcmp(what, "all");
  const int n = 235868, d = 128;
  float *x, *y, *big, *out;
  const size_t big_bytes = (size_t)2 << 30;
  CK(hipMalloc(&x, (size_t)n * d * 4)); CK(hipMalloc(&y, (size_t)n * d * 4));
  CK(hipMalloc(&big, big_bytes)); CK(hipMalloc(&out, 256));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, x, (size_t)n * d);
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, big, big_bytes / 4);
  CK(hipDeviceSynchronize());

  if (all || !strcmp(what, "stream")) {
    printf("# stream: coalesced 16-B/lane reads, grid 2048 x 256, U loads in flight per lane\n");
    for (size_t mb : {2, 16, 64, 128, 192, 512, 2048})
      for (int shared : {0, 1})
        for (int U : {4, 8}) {
          if (shared && mb > 192) continue;
          const size_t n4 = mb * 1024 * 1024 / 16;
          auto launch = [&]() {
            if (U == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(2048), dim3(256), 0, 0, (const float4*)big, n4, shared, out);
            else hipLaunchKernelGGL(stream_kernel<8>, dim3(2048), dim3(256), 0, 0, (const float4*)big, n4, shared, out);
          };
          const double us = time_us(launch);
          const double bytes = (double)mb * 1048576.0 * (shared ? 8 : 1);
          printf("stream buf=%zu MB %s U=%d: %.1f us  %.2f TB/s (bytes requested by the CUs)\n", mb,
                 shared ? "every XCD reads all of it" : "one eighth per XCD", U, us, bytes / us / 1e6);
        }
  }
  if (all || !strcmp(what, "gather")) {
    printf("# gather: TL local + TR remote trips per output row (2 rows of 512 B per trip), n=%d rows written\n", n);
    const int mixes[][2] = {{4, 0}, {3, 1}, {2, 2}, {1, 3}, {0, 4}, {0, 2}, {2, 0}, {0, 8}, {8, 0}, {5, 3}, {3, 2}};
    for (auto& m : mixes)
      for (int U : {4}) {
        auto launch = [&]() {
          hipLaunchKernelGGL(gather_kernel<4>, dim3(2048), dim3(256), 0, 0, x, y, n, m[0], m[1], 0xff, 8);
        };
        const double us = time_us(launch);
        const double rows = (double)n * 2 * (m[0] + m[1]);
        printf("gather TL=%d TR=%d U=%d: %.1f us  %.1f ps per gathered row  (%.2f TB/s gathered, remote %.2f TB/s)\n", m[0],
               m[1], U, us, us * 1e6 / rows, rows * 512 / us / 1e6, (double)n * 2 * m[1] * 512 / us / 1e6);
      }
    printf("# remote-only gather (TR=4) on a subset of the XCDs: per-XCD or chip-wide limit?\n");
    for (int mask : {0xff, 0x0f, 0x55, 0x03, 0x01}) {
      const int na = __builtin_popcount(mask);
      auto launch = [&]() {
        hipLaunchKernelGGL(gather_kernel<4>, dim3(2048), dim3(256), 0, 0, x, y, n, 0, 4, mask, na);
      };
      const double us = time_us(launch);
      const double bytes = (double)n * 8 * 512;
      printf("gather remote-only xcd_mask=0x%02x (%d XCDs): %.1f us  %.2f TB/s  = %.2f TB/s per active XCD\n", mask, na, us,
             bytes / us / 1e6, bytes / us / 1e6 / na);
    }
    printf("# remote-only gather by grid size (resident waves per CU)\n");
    for (int grid : {256, 512, 1024, 2048, 4096})
      for (int U : {4, 8}) {
        auto launch = [&]() {
          if (U == 4) hipLaunchKernelGGL(gather_kernel<4>, dim3(grid), dim3(256), 0, 0, x, y, n, 0, 8, 0xff, 8);
          else hipLaunchKernelGGL(gather_kernel<8>, dim3(grid), dim3(256), 0, 0, x, y, n, 0, 8, 0xff, 8);
        };
        const double us = time_us(launch);
        printf("gather remote-only TR=8 grid=%d (%d waves/CU) U=%d: %.1f us  %.2f TB/s\n", grid, grid * 4 / 256 > 32 ? 32 : grid * 4 / 256,
               U, us, (double)n * 16 * 512 / us / 1e6);
      }
  }
  if (all || !strcmp(what, "ldswin")) {
    printf("# ldswin: the block's row window staged in LDS, local trips by ds_read_b128, remote trips from global\n");
    const int mixes[][2] = {{4, 0}, {3, 1}, {2, 2}, {1, 3}, {0, 4}, {5, 3}, {3, 2}, {0, 0}};
    for (auto& m : mixes) {
      double us;
#define RUN(WR, NT, DMA, name)                                                                                          \
  {                                                                                                                     \
    const int nwin = (n + WR - 1) / WR;                                                                                 \
    CK(hipFuncSetAttribute((const void*)ldswin_kernel<WR, NT, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, WR * 512)); \
    auto launch = [&]() {                                                                                               \
      hipLaunchKernelGGL((ldswin_kernel<WR, NT, DMA>), dim3(nwin), dim3(NT), WR * 512, 0, x, y, n, m[0], m[1]);          \
    };                                                                                                                  \
    us = time_us(launch);                                                                                               \
    printf("ldswin %s TL=%d TR=%d: %.1f us  (%.1f ps per gathered row)\n", name, m[0], m[1], us,                        \
           m[0] + m[1] ? us * 1e6 / ((double)n * 2 * (m[0] + m[1])) : 0.0);                                             \
  }
      RUN(128, 512, false, "window 128 rows / 512 thr (2 blocks per CU), ds_write");
      RUN(128, 512, true, "window 128 rows / 512 thr (2 blocks per CU), LDS-DMA");
      RUN(64, 256, false, "window  64 rows / 256 thr (5 blocks per CU), ds_write");
      RUN(64, 256, true, "window  64 rows / 256 thr (5 blocks per CU), LDS-DMA");
      RUN(256, 1024, false, "window 256 rows / 1024 thr (1 block per CU), ds_write");
      RUN(256, 1024, true, "window 256 rows / 1024 thr (1 block per CU), LDS-DMA");
      RUN(128, 1024, true, "window 128 rows / 1024 thr (2 blocks per CU), LDS-DMA");
#undef RUN
    }
  }
  return 0;
}
