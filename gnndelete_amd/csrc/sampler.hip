// Random walks for GraphSAINT mini-batches (torch_geometric's GraphSAINTRandomWalkSampler -> torch_sparse
// random_walk, used at framework/trainer/gnndelete_nodeemb.py:379-381 and :734-736 with walk_length 2):
// walker w starts at start[w] and moves walk_length times to a uniformly drawn out-neighbour of its current node
// (a node without out-edges keeps the walker).  One thread per walker; the draw for (walker, step) is a
// counter-based hash of (seed, walker, step) - no generator state, any launch geometry gives the same walks.
#include "common.h"

namespace gd {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {     // splitmix64 finaliser
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void random_walk_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                          const int64_t* __restrict__ start, int32_t n_walks,
                                                          int32_t walk_length, uint64_t seed, int64_t* __restrict__ out) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= n_walks) return;
  int32_t cur = (int32_t)start[w];
  out[w] = cur;
  for (int s = 1; s <= walk_length; ++s) {
    const int32_t r0 = rowptr[cur], deg = rowptr[cur + 1] - r0;
    if (deg > 0) {
      const uint64_t u = mix64(mix64(seed ^ ((uint64_t)w << 20)) + (uint64_t)s);
      // multiply-shift maps the 32 high bits to [0, deg) without the modulo bias of u % deg
      cur = col[r0 + (int32_t)(((u >> 32) * (uint64_t)deg) >> 32)];
    }
    out[(int64_t)s * n_walks + w] = cur;
  }
}

}  // namespace gd

extern "C" int gd_random_walk(const int32_t* rowptr, const int32_t* col, int32_t n_nodes, const int64_t* start,
                              int32_t n_walks, int32_t walk_length, uint64_t seed, int64_t* out, void* stream) {
  using namespace gd;
  GD_REQUIRE(rowptr && start && out && (col || n_walks == 0), GD_E_NULL, "gd_random_walk: null pointer");
  GD_REQUIRE(n_nodes > 0 && n_walks >= 0 && walk_length >= 0, GD_E_DIM, "gd_random_walk: bad sizes");
  if (n_walks == 0) return GD_OK;
  hipLaunchKernelGGL(random_walk_kernel, dim3((n_walks + 255) / 256), dim3(256), 0, (hipStream_t)stream, rowptr, col, start,
                     n_walks, walk_length, seed, out);
  return launched("random_walk");
}
