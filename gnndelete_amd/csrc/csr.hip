// COO -> CSR on the device: the layout step in front of every message-passing kernel.
//
// The reference hands torch_geometric a [2, E] int64 edge_index on every call and PyG re-derives its
// scatter indices from it each time (framework/models/gcn.py:16-22, gat.py:16-22, gin.py:26-34); the Df /
// S_Df preprocessing sorts and coalesces the same list with to_undirected (delete_gnn.py:175-182).  Here the
// list is turned ONCE into a CSR over TARGET rows with the sources of a row ascending - the order every kernel
// of this library sums in, so results do not depend on the order the edges arrive in.
//
// Integer work only, bit-exact and deterministic: key = dst * n + src (< 2^62), a STABLE least-significant-
// digit radix sort of (key, input position) over exactly the bits n^2 needs (rocPRIM's device radix sort - a
// library primitive, like rocBLAS for a plain GEMM), then col = key mod n and rowptr[i] = lower_bound(keys,
// i * n) by binary search (no atomics, no dependence on arrival order).  Multi-edges are kept (PyG's
// convolutions count them), ties keep their input order.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace gd {

__global__ __launch_bounds__(256) void coo_keys_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                       int64_t n_edges, int64_t n, uint64_t* __restrict__ keys,
                                                       uint32_t* __restrict__ pos, int32_t* __restrict__ bad) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_edges) return;
  const int64_t s = src[e], d = dst[e];
  if (s < 0 || s >= n || d < 0 || d >= n) {
    *bad = 1;                                       // every offender writes the same value
    keys[e] = ~0ull;
  } else {
    keys[e] = (uint64_t)d * (uint64_t)n + (uint64_t)s;
  }
  pos[e] = (uint32_t)e;
}

__global__ __launch_bounds__(256) void csr_cols_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ pos,
                                                       int64_t n_edges, int64_t n, int32_t* __restrict__ col,
                                                       int32_t* __restrict__ order) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n_edges) return;
  col[k] = (int32_t)(keys[k] % (uint64_t)n);
  if (order) order[k] = (int32_t)pos[k];
}

__global__ __launch_bounds__(256) void csr_rowptr_kernel(const uint64_t* __restrict__ keys, int64_t n_edges, int64_t n,
                                                         int32_t* __restrict__ rowptr) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i > n) return;
  const uint64_t first = (uint64_t)i * (uint64_t)n;      // smallest key of row i (row n: one past the last)
  int64_t lo = 0, hi = n_edges;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < first) lo = mid + 1;
    else hi = mid;
  }
  rowptr[i] = (int32_t)lo;
}

static unsigned key_bits(int64_t n) {
  unsigned b = 1;
  const unsigned __int128 top = (unsigned __int128)n * (unsigned __int128)n;   // keys are < n^2
  while (b < 64 && ((unsigned __int128)1 << b) < top) ++b;
  return b;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct CooLayout {
  size_t keys_in, keys_out, pos_in, pos_out, flag, temp, temp_bytes, total;
};

static hipError_t coo_layout(int64_t n_edges, int64_t n, CooLayout* L) {
  size_t temp = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, temp, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                                           (uint32_t*)nullptr, (size_t)n_edges, 0u, key_bits(n), (hipStream_t)0);
  if (e != hipSuccess) return e;
  size_t off = 0;
  L->keys_in = off; off += align256((size_t)n_edges * 8);
  L->keys_out = off; off += align256((size_t)n_edges * 8);
  L->pos_in = off; off += align256((size_t)n_edges * 4);
  L->pos_out = off; off += align256((size_t)n_edges * 4);
  L->flag = off; off += 256;
  L->temp = off; off += align256(temp);
  L->temp_bytes = temp;
  L->total = off;
  return hipSuccess;
}

}  // namespace gd

extern "C" int64_t gd_csr_from_coo_workspace(int32_t n_nodes, int64_t n_edges) {
  using namespace gd;
  if (n_nodes < 0 || n_edges < 0 || n_edges >= (1ll << 31)) return -1;
  if (n_edges == 0 || n_nodes == 0) return 256;
  CooLayout L;
  if (coo_layout(n_edges, n_nodes, &L) != hipSuccess) return -1;
  return (int64_t)L.total;
}

extern "C" int gd_csr_from_coo(const int64_t* src, const int64_t* dst, int64_t n_edges, int32_t n_nodes, int32_t* rowptr,
                               int32_t* col, int32_t* order, int32_t* status, void* workspace, int64_t workspace_bytes,
                               void* stream) {
  using namespace gd;
  GD_REQUIRE(rowptr && status, GD_E_NULL, "gd_csr_from_coo: null rowptr / status");
  GD_REQUIRE(n_nodes >= 0 && n_edges >= 0 && n_edges < (1ll << 31), GD_E_DIM,
             "gd_csr_from_coo: n_nodes=%d n_edges=%lld outside the int32 CSR range", n_nodes, (long long)n_edges);
  GD_REQUIRE(n_edges == 0 || (src && dst && col && workspace), GD_E_NULL, "gd_csr_from_coo: null pointer");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(status, 0, sizeof(int32_t), s);
  if (e != hipSuccess) return fail(-(int)e, "gd_csr_from_coo: %s", hipGetErrorString(e));
  if (n_edges == 0 || n_nodes == 0) {
    e = hipMemsetAsync(rowptr, 0, ((size_t)n_nodes + 1) * sizeof(int32_t), s);
    if (e != hipSuccess) return fail(-(int)e, "gd_csr_from_coo: %s", hipGetErrorString(e));
    if (n_edges > 0) {                               // edges without nodes: every endpoint is out of range
      const int32_t one = 1;
      e = hipMemcpyAsync(status, &one, sizeof(one), hipMemcpyHostToDevice, s);
      if (e != hipSuccess) return fail(-(int)e, "gd_csr_from_coo: %s", hipGetErrorString(e));
    }
    return GD_OK;
  }
  CooLayout L;
  e = coo_layout(n_edges, n_nodes, &L);
  if (e != hipSuccess) return fail(-(int)e, "gd_csr_from_coo: %s", hipGetErrorString(e));
  GD_REQUIRE(workspace_bytes >= (int64_t)L.total, GD_E_WORKSPACE, "gd_csr_from_coo: workspace %lld < %lld bytes",
             (long long)workspace_bytes, (long long)L.total);
  GD_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, GD_E_ALIGN, "gd_csr_from_coo: workspace not 256-byte aligned");
  char* ws = reinterpret_cast<char*>(workspace);
  uint64_t* keys_in = reinterpret_cast<uint64_t*>(ws + L.keys_in);
  uint64_t* keys_out = reinterpret_cast<uint64_t*>(ws + L.keys_out);
  uint32_t* pos_in = reinterpret_cast<uint32_t*>(ws + L.pos_in);
  uint32_t* pos_out = reinterpret_cast<uint32_t*>(ws + L.pos_out);
  const dim3 block(256), grid_e((unsigned)((n_edges + 255) / 256)), grid_n((unsigned)(((int64_t)n_nodes + 1 + 255) / 256));
  hipLaunchKernelGGL(coo_keys_kernel, grid_e, block, 0, s, src, dst, n_edges, (int64_t)n_nodes, keys_in, pos_in, status);
  size_t temp = L.temp_bytes;
  e = rocprim::radix_sort_pairs(ws + L.temp, temp, keys_in, keys_out, pos_in, pos_out, (size_t)n_edges, 0u,
                                key_bits(n_nodes), s);
  if (e != hipSuccess) return fail(-(int)e, "gd_csr_from_coo: radix sort: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(csr_cols_kernel, grid_e, block, 0, s, keys_out, pos_out, n_edges, (int64_t)n_nodes, col, order);
  hipLaunchKernelGGL(csr_rowptr_kernel, grid_n, block, 0, s, keys_out, n_edges, (int64_t)n_nodes, rowptr);
  return launched("csr_from_coo");
}
