// Stand-alone lab for the balanced SpMM on bench.py's step graph (tools/experiments/bench_graph.bin, written by
// spmm_traffic_floor.py-style code on the CPU: the synth-collab request in the engine's locality order).
// Links the PRODUCTION kernels (csrc/spmm.hip is included as source) so that variants are compared against exactly
// what the library runs.  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -I../../gnndelete_amd/csrc
#include "../../gnndelete_amd/csrc/spmm.hip"

#include <algorithm>
#include <string>
#include <vector>

namespace gd { char* error_buffer() { static thread_local char b[256]; return b; } }

// the production body + a wall-clock stamp (100 MHz) per block: when does each XCD finish its range?
template <int LPR, int U>
__global__ __launch_bounds__(256) void stamped_kernel(const int4* items, int32_t n_items, const int32_t* xcd_bounds,
                                                      const int32_t* col, const float* val, const float* x, int64_t ldx,
                                                      float* y, int64_t ldy, const float* bias, float* scratch, int32_t d4,
                                                      int32_t nnz, uint64_t* stamps) {
  const uint64_t t0 = wall_clock64();
  gd::spmm_persist_body<LPR, 1, U, true, true>(items, n_items, xcd_bounds, col, val, x, ldx, y, ldy, bias, 0.f, x, scratch, d4,
                                               nnz);
  __syncthreads();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = wall_clock64(); }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <typename T> static T* dev(const std::vector<T>& v) {
  T* p; CK(hipMalloc(&p, std::max<size_t>(v.size(), 4) * sizeof(T)));
  CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}

template <typename F> static double time_us(F launch, int reps = 30) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / reps * 1e3;
}

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "tools/experiments/bench_graph.bin";
  FILE* f = fopen(path, "rb");
  if (!f) { printf("cannot open %s\n", path); return 1; }
  int32_t hdr[2]; fread(hdr, 4, 2, f);
  const int n = hdr[0], nnz = hdr[1];
  std::vector<int32_t> rowptr(n + 1), col(nnz);
  fread(rowptr.data(), 4, n + 1, f); fread(col.data(), 4, nnz, f); fclose(f);
  std::vector<float> val(nnz);
  for (int i = 0; i < n; ++i)
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k)
      val[k] = 1.0f / sqrtf((float)(rowptr[i + 1] - rowptr[i])) / sqrtf((float)(rowptr[col[k] + 1] - rowptr[col[k]]));
  // work items exactly like gnndelete_amd/graph.py:SplitPlan
  std::vector<int32_t> items, split; int n_slots = 0;
  std::vector<double> item_cost;
  for (int i = 0; i < n; ++i) {
    const int s = rowptr[i], e = rowptr[i + 1], pieces = std::max(1, (e - s + 63) / 64);
    if (pieces > 1) { split.insert(split.end(), {i, n_slots, pieces, 0}); }
    for (int p = 0; p < pieces; ++p) {
      const int a = s + p * 64, b = std::min(e, a + 64);
      items.insert(items.end(), {i, a, b, pieces > 1 ? n_slots++ : -1});
      item_cost.push_back((b - a) + 1.0);
    }
  }
  const int n_items = items.size() / 4, n_split = split.size() / 4;
  printf("graph: n=%d nnz=%d items=%d split rows=%d slots=%d\n", n, nnz, n_items, n_split, n_slots);
  int32_t *d_items = dev(items), *d_split = dev(split), *d_col = dev(col);
  float* d_val = dev(val);
  for (int d : {128, 64}) {
    std::vector<float> hx((size_t)n * d), hb(d);
    uint32_t s = 12345;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    for (auto& v : hb) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    float *x = dev(hx), *b = dev(hb), *y, *y2, *scratch;
    CK(hipMalloc(&y, (size_t)n * d * 4)); CK(hipMalloc(&y2, (size_t)n * d * 4));
    CK(hipMalloc(&scratch, (size_t)std::max(n_slots, 1) * d * 4));
    const double alg = 4.0 * (n + 1) + 8.0 * nnz + 8.0 * n * d;
    auto prod = [&]() {
      int rc = gd_spmm_csr_balanced_f32(d_items, n_items, d_split, n_split, d_col, d_val, x, d, y, d, b, 0.f, nullptr, scratch, d,
                                        nnz, n, nullptr, nullptr);
      if (rc) { printf("rc=%d %s\n", rc, gd::error_buffer()); exit(1); }
    };
    const double us = time_us(prod);
    printf("d=%d production (persist + fix-up): %.1f us  %.0f GB/s algorithmic = %.3f of 8 TB/s\n", d, us, alg / us / 1e3, alg / us / 8e6);
    // reference on the host for the first 2000 rows + the split rows
    std::vector<float> hy((size_t)n * d);
    CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (int i = 0; i < n; i += (i < 2000 ? 1 : 97)) {
      for (int c = 0; c < d; ++c) {
        double acc = hb[c];
        for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) acc += (double)val[k] * hx[(size_t)col[k] * d + c];
        const double diff = acc - hy[(size_t)i * d + c];
        num += diff * diff; den += acc * acc;
      }
    }
    printf("d=%d production rel err vs fp64 host (sampled rows): %.2e\n", d, sqrt(num / den));
    // --- XCD ranges balanced by rows moved
    std::vector<int32_t> bounds(9, 0);
    {
      double tot = 0; for (double c : item_cost) tot += c;
      double acc = 0; int k = 1;
      for (int i = 0; i < n_items && k < 8; ++i) { acc += item_cost[i]; while (k < 8 && acc >= tot * k / 8) bounds[k++] = i + 1; }
      bounds[8] = n_items;
    }
    int32_t* d_bounds = dev(bounds);
    printf("balanced bounds:"); for (int b : bounds) printf(" %d", b); printf("\n");
    auto bal = [&]() {
      int rc = gd_spmm_csr_balanced_f32(d_items, n_items, d_split, n_split, d_col, d_val, x, d, y2, d, b, 0.f, nullptr, scratch, d,
                                        nnz, n, d_bounds, nullptr);
      if (rc) { printf("rc=%d %s\n", rc, gd::error_buffer()); exit(1); }
    };
    const double usb = time_us(bal);
    printf("d=%d balanced XCD ranges (persist + fix-up): %.1f us  = %.3f of 8 TB/s\n", d, usb, alg / usb / 8e6);
    {
      std::vector<float> hy2((size_t)n * d);
      CK(hipMemcpy(hy2.data(), y2, hy2.size() * 4, hipMemcpyDeviceToHost));
      size_t bad = 0; for (size_t i = 0; i < hy.size(); ++i) bad += hy[i] != hy2[i];
      printf("d=%d balanced vs production: %zu differing floats (must be 0)\n", d, bad);
    }
    // --- per-XCD finish times and grid sweep (persist kernel only, no fix-up)
    uint64_t* stamps; CK(hipMalloc(&stamps, 2 * 16384 * 8));
    for (int a : {-1, 1, 6, 11, 16, 24, 40}) for (int grid : {2048, 8192}) {
      const int use_b = a >= 0;
      if (use_b) {
        std::vector<int32_t> bb(9, 0);
        double tot = 0; for (double c : item_cost) tot += c - 1.0 + a;
        double acc = 0; int k = 1;
        for (int i = 0; i < n_items && k < 8; ++i) { acc += item_cost[i] - 1.0 + a; while (k < 8 && acc >= tot * k / 8) bb[k++] = i + 1; }
        bb[8] = n_items;
        CK(hipMemcpy(d_bounds, bb.data(), 36, hipMemcpyHostToDevice));
      }
      auto launch = [&]() {
        if (d == 128) hipLaunchKernelGGL((stamped_kernel<32, 4>), dim3(grid), dim3(256), 0, 0, (const int4*)d_items, n_items,
                                         use_b ? d_bounds : nullptr, d_col, d_val, x, (int64_t)d, y2, (int64_t)d, b, scratch, d / 4, nnz, stamps);
        else hipLaunchKernelGGL((stamped_kernel<16, 4>), dim3(grid), dim3(256), 0, 0, (const int4*)d_items, n_items,
                                use_b ? d_bounds : nullptr, d_col, d_val, x, (int64_t)d, y2, (int64_t)d, b, scratch, d / 4, nnz, stamps);
      };
      const double t = time_us(launch);
      std::vector<uint64_t> hs(2 * grid);
      CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
      uint64_t t0 = ~0ull; for (int bI = 0; bI < grid; ++bI) t0 = std::min(t0, hs[2 * bI]);
      printf("d=%d cost=edges+%d grid=%d: %.1f us; XCD finish (us after first start):", d, a, grid, t);
      for (int xcd = 0; xcd < 8; ++xcd) {
        uint64_t e = 0; for (int bI = xcd; bI < grid; bI += 8) e = std::max(e, hs[2 * bI + 1]);
        printf(" %.1f", (double)(e - t0) / 100.0);
      }
      printf("\n");
    }
    CK(hipFree(stamps)); CK(hipFree(d_bounds));
    CK(hipFree(x)); CK(hipFree(b)); CK(hipFree(y)); CK(hipFree(y2)); CK(hipFree(scratch));
  }
  return 0;
}
