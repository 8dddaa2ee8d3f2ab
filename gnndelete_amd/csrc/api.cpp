// ABI bookkeeping: version + thread-local error string.
#include "common.h"

#include <stdlib.h>

namespace gd {
char* error_buffer() {
  static thread_local char buf[256] = "";
  return buf;
}

// How the dense fp32 products are formed (rows_gemm.hip): 0 = v_mfma_f32_32x32x2_f32, 6 = split arithmetic on the
// bf16 matrix instruction.  Process-wide; GD_MATRIX_SPLIT in the environment sets the initial value.
static int g_matrix_split = [] {
  const char* e = getenv("GD_MATRIX_SPLIT");
  const int v = e ? atoi(e) : GD_MATRIX_SPLIT_DEFAULT;
  return v == 6 ? 6 : 0;
}();
int matrix_split() { return g_matrix_split; }
}  // namespace gd

#include "build_stamp.inc"
extern "C" int gd_abi_version(void) { return GD_ABI_VERSION; }
extern "C" const char* gd_build_source_hash(void) { return GD_BUILD_SOURCE_HASH; }
extern "C" const char* gd_last_error_string(void) { return gd::error_buffer(); }
extern "C" int gd_matrix_split(void) { return gd::g_matrix_split; }
extern "C" int gd_set_matrix_split(int n_products) {
  GD_REQUIRE(n_products == 0 || n_products == 6, GD_E_DIM, "gd_set_matrix_split: 0 or 6 (got %d)", n_products);
  gd::g_matrix_split = n_products;
  return GD_OK;
}
