// RCCL entry points of the C ABI (gd_allreduce_f32 and the row exchange of the partitioned step).
// RCCL is resolved at first use with dlsym - first among the libraries already in the process (a PyTorch-ROCm process has
// its own librccl mapped), then librccl.so by name - so the library has no link-time dependency on it and loads, exports
// and validates its arguments on a box without a GPU or without RCCL.
#include <dlfcn.h>

#include "common.h"

namespace {
struct UniqueId { char bytes[128]; };                      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                                        // ncclComm_t
constexpr int kNcclFloat32 = 7, kNcclSum = 0;              // ncclDataType_t / ncclRedOp_t values of nccl.h

struct Rccl {
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl t;
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclAllReduce")) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h && !(h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL))) return t;
#define GD_SYM(field, name) t.field = reinterpret_cast<decltype(t.field)>(dlsym(h, name))
    GD_SYM(GetUniqueId, "ncclGetUniqueId");
    GD_SYM(CommInitRank, "ncclCommInitRank");
    GD_SYM(CommDestroy, "ncclCommDestroy");
    GD_SYM(AllReduce, "ncclAllReduce");
    GD_SYM(Send, "ncclSend");
    GD_SYM(Recv, "ncclRecv");
    GD_SYM(GroupStart, "ncclGroupStart");
    GD_SYM(GroupEnd, "ncclGroupEnd");
    GD_SYM(GetErrorString, "ncclGetErrorString");
#undef GD_SYM
    t.ok = t.GetUniqueId && t.CommInitRank && t.CommDestroy && t.AllReduce && t.Send && t.Recv && t.GroupStart && t.GroupEnd;
    return t;
  }();
  return r;
}

int rccl_fail(const char* what, int rc) {
  const Rccl& r = rccl();
  return gd::fail(-1000 - rc, "%s: RCCL error %d (%s)", what, rc, r.GetErrorString ? r.GetErrorString(rc) : "?");
}
}  // namespace

#define GD_NEED_RCCL(what) GD_REQUIRE(rccl().ok, GD_E_NULL, what ": RCCL (librccl.so) is not available in this process")

extern "C" int gd_comm_unique_id(void* id128) {
  GD_REQUIRE(id128, GD_E_NULL, "gd_comm_unique_id: null pointer");
  GD_NEED_RCCL("gd_comm_unique_id");
  const int rc = rccl().GetUniqueId(reinterpret_cast<UniqueId*>(id128));
  return rc ? rccl_fail("gd_comm_unique_id", rc) : GD_OK;
}

extern "C" int gd_comm_init(const void* id128, int32_t n_ranks, int32_t rank, void** comm) {
  GD_REQUIRE(id128 && comm, GD_E_NULL, "gd_comm_init: null pointer");
  GD_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, GD_E_DIM, "gd_comm_init: rank %d of %d", rank, n_ranks);
  GD_NEED_RCCL("gd_comm_init");
  UniqueId id = *reinterpret_cast<const UniqueId*>(id128);
  Comm c = nullptr;
  const int rc = rccl().CommInitRank(&c, n_ranks, id, rank);
  if (rc) return rccl_fail("gd_comm_init", rc);
  *comm = c;
  return GD_OK;
}

extern "C" int gd_comm_destroy(void* comm) {
  if (!comm) return GD_OK;
  GD_NEED_RCCL("gd_comm_destroy");
  const int rc = rccl().CommDestroy(comm);
  return rc ? rccl_fail("gd_comm_destroy", rc) : GD_OK;
}

extern "C" int gd_allreduce_f32(float* buf, int64_t n, void* comm, void* stream) {
  GD_REQUIRE(buf && comm, GD_E_NULL, "gd_allreduce_f32: null pointer");
  GD_REQUIRE(n >= 0, GD_E_DIM, "gd_allreduce_f32: n < 0");
  if (n == 0) return GD_OK;
  GD_NEED_RCCL("gd_allreduce_f32");
  const int rc = rccl().AllReduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, comm, (hipStream_t)stream);
  return rc ? rccl_fail("gd_allreduce_f32", rc) : GD_OK;
}

extern "C" int gd_exchange_rows_f32(const float* send, const int64_t* send_rows, float* recv, const int64_t* recv_rows,
                                    int32_t row_elems, int32_t n_ranks, void* comm, void* stream) {
  GD_REQUIRE(send_rows && recv_rows && comm, GD_E_NULL, "gd_exchange_rows_f32: null pointer");
  GD_REQUIRE(row_elems > 0 && n_ranks >= 1, GD_E_DIM, "gd_exchange_rows_f32: bad dims");
  int64_t ns = 0, nr = 0;
  for (int p = 0; p < n_ranks; ++p) {
    GD_REQUIRE(send_rows[p] >= 0 && recv_rows[p] >= 0, GD_E_DIM, "gd_exchange_rows_f32: negative row count for peer %d", p);
    ns += send_rows[p];
    nr += recv_rows[p];
  }
  GD_REQUIRE((ns == 0 || send) && (nr == 0 || recv), GD_E_NULL, "gd_exchange_rows_f32: null buffer");
  if (ns == 0 && nr == 0) return GD_OK;
  GD_NEED_RCCL("gd_exchange_rows_f32");
  const Rccl& r = rccl();
  int rc = r.GroupStart();
  if (rc) return rccl_fail("gd_exchange_rows_f32 (group start)", rc);
  int64_t so = 0, ro = 0;
  for (int p = 0; p < n_ranks && !rc; ++p) {
    if (send_rows[p]) rc = r.Send(send + so * row_elems, (size_t)(send_rows[p] * row_elems), kNcclFloat32, p, comm, (hipStream_t)stream);
    if (!rc && recv_rows[p]) rc = r.Recv(recv + ro * row_elems, (size_t)(recv_rows[p] * row_elems), kNcclFloat32, p, comm, (hipStream_t)stream);
    so += send_rows[p];
    ro += recv_rows[p];
  }
  const int rc_end = r.GroupEnd();
  if (rc) return rccl_fail("gd_exchange_rows_f32", rc);
  return rc_end ? rccl_fail("gd_exchange_rows_f32 (group end)", rc_end) : GD_OK;
}
