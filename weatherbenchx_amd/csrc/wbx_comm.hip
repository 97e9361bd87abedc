// The cross-rank combine behind the C ABI (include/wbx.h, "accumulators across ranks"): a rank's accumulator buffer is summed
// over the ranks in place with ONE ncclAllReduce(sum, double) on the context's stream -- RCCL over xGMI, one process per GPU --
// the counterpart of beam.CombinePerKey(CombiningSum()) (beam_pipeline.py:509-510, beam_utils.py:30-50).
//
// RCCL is bound at first use with dlopen / dlsym, not at link time: libwbx_hip.so keeps loading on hosts without RCCL, and
// inside a process that already holds a librccl.so.1 (PyTorch ships its own) the SAME library is reused -- two RCCL copies
// in one process would fight over global symbols.  WBX_RCCL_PATH overrides the search.
#include <dlfcn.h>

#include <cstdlib>
#include <mutex>

#include "wbx_common.hpp"

namespace wbx {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2.x: rccl.h:40-43, 187, 220, 260, 339, 448-467, 611)
struct RcclUniqueId {
  char internal[WBX_COMM_ID_BYTES];
};
typedef struct ncclComm* RcclComm;
typedef int (*fn_get_unique_id)(RcclUniqueId*);
typedef int (*fn_comm_init_rank)(RcclComm*, int, RcclUniqueId, int);
typedef int (*fn_comm_destroy)(RcclComm);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int /*dtype*/, int /*op*/, RcclComm, hipStream_t);
typedef const char* (*fn_error_string)(int);
constexpr int RCCL_FLOAT64 = 8, RCCL_SUM = 0;

struct Rccl {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_error_string error_string = nullptr;
  char why[256] = {0};
};

static Rccl& rccl_state() {
  static Rccl r;
  return r;
}

static Rccl* rccl() {
  Rccl& r = rccl_state();
  static std::once_flag once;
  std::call_once(once, [&r] {
    const char* names[] = {getenv("WBX_RCCL_PATH"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names) {
      if (n == nullptr || *n == 0) continue;
      r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
      snprintf(r.why, sizeof(r.why), "%s", dlerror());
    }
    if (!r.handle) return;
    r.get_unique_id = reinterpret_cast<fn_get_unique_id>(dlsym(r.handle, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<fn_comm_init_rank>(dlsym(r.handle, "ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<fn_comm_destroy>(dlsym(r.handle, "ncclCommDestroy"));
    r.all_reduce = reinterpret_cast<fn_all_reduce>(dlsym(r.handle, "ncclAllReduce"));
    r.error_string = reinterpret_cast<fn_error_string>(dlsym(r.handle, "ncclGetErrorString"));
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.error_string) {
      snprintf(r.why, sizeof(r.why), "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
      dlclose(r.handle);
      r.handle = nullptr;
    }
  });
  return r.handle ? &r : nullptr;
}

static const char* rccl_why() {
  (void)rccl();
  static char msg[384];
  const char* why = rccl_state().why;
  snprintf(msg, sizeof(msg), "librccl.so.1 could not be loaded (set WBX_RCCL_PATH)%s%s", *why ? ": " : "", why);
  return msg;
}

}  // namespace wbx

struct wbx_comm {
  wbx::RcclComm comm = nullptr;
  int nranks = 1, rank = 0, device = 0;
  int64_t collectives = 0;
};

#define WBX_RCCL(expr)                                                                              \
  do {                                                                                              \
    int _e = (expr);                                                                                \
    if (_e != 0) return wbx::fail(WBX_ERR_RCCL, "%s failed: %s", #expr, R->error_string(_e));       \
  } while (0)

extern "C" int wbx_comm_unique_id(void* id_out) {
  using namespace wbx;
  WBX_REQUIRE(id_out != nullptr, "id_out is NULL");
  Rccl* R = rccl();
  if (!R) return fail(WBX_ERR_RCCL, "%s", rccl_why());
  RcclUniqueId id;
  WBX_RCCL(R->get_unique_id(&id));
  memcpy(id_out, id.internal, WBX_COMM_ID_BYTES);
  return 0;
}

extern "C" int wbx_comm_create(wbx_ctx* ctx, const void* unique_id, int32_t nranks, int32_t rank, wbx_comm** out) {
  using namespace wbx;
  WBX_REQUIRE(ctx != nullptr && unique_id != nullptr && out != nullptr, "NULL argument");
  WBX_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d outside a group of %d", rank, nranks);
  Rccl* R = rccl();
  if (!R) return fail(WBX_ERR_RCCL, "%s", rccl_why());
  WBX_HIP(hipSetDevice(ctx->device));
  RcclUniqueId id;
  memcpy(id.internal, unique_id, WBX_COMM_ID_BYTES);
  wbx_comm* c = new wbx_comm();
  c->nranks = nranks;
  c->rank = rank;
  c->device = ctx->device;
  int e = R->comm_init_rank(&c->comm, nranks, id, rank);
  if (e != 0) {
    delete c;
    return fail(WBX_ERR_RCCL, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, R->error_string(e));
  }
  *out = c;
  return 0;
}

extern "C" int wbx_comm_destroy(wbx_comm* comm) {
  using namespace wbx;
  if (comm == nullptr) return 0;
  Rccl* R = rccl();
  if (R && comm->comm) {
    (void)hipSetDevice(comm->device);
    R->comm_destroy(comm->comm);
  }
  delete comm;
  return 0;
}

extern "C" int wbx_comm_info(const wbx_comm* comm, int32_t* nranks_out, int32_t* rank_out, int64_t* collectives_out) {
  WBX_REQUIRE(comm != nullptr, "comm is NULL");
  if (nranks_out) *nranks_out = comm->nranks;
  if (rank_out) *rank_out = comm->rank;
  if (collectives_out) *collectives_out = comm->collectives;
  return 0;
}

extern "C" int wbx_acc_allreduce(wbx_ctx* ctx, wbx_comm* comm, double* acc, int64_t n) {
  using namespace wbx;
  WBX_REQUIRE(ctx != nullptr && comm != nullptr, "NULL argument");
  WBX_REQUIRE(n >= 0, "n must be >= 0");
  WBX_REQUIRE(ctx->device == comm->device, "the communicator was created on device %d, the context is on device %d", comm->device,
              ctx->device);
  if (n == 0) return 0;  // (every rank passes the same n: an empty layout is empty everywhere)
  WBX_REQUIRE(acc != nullptr, "acc is NULL");
  Rccl* R = rccl();
  if (!R) return fail(WBX_ERR_RCCL, "%s", rccl_why());
  WBX_HIP(hipSetDevice(ctx->device));
  WBX_RCCL(R->all_reduce(acc, acc, (size_t)n, RCCL_FLOAT64, RCCL_SUM, comm->comm, ctx->stream));
  comm->collectives += 1;
  return 0;
}

extern "C" int wbx_acc_read(wbx_ctx* ctx, const double* acc, int64_t n, double* host_out) {
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  WBX_REQUIRE(n >= 0, "n must be >= 0");
  if (n == 0) return 0;
  WBX_REQUIRE(acc != nullptr && host_out != nullptr, "NULL pointer");
  WBX_HIP(hipSetDevice(ctx->device));
  WBX_HIP(hipMemcpyAsync(host_out, acc, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  WBX_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int wbx_acc_reset(wbx_ctx* ctx, double* acc, int64_t n) {
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  WBX_REQUIRE(n >= 0, "n must be >= 0");
  if (n == 0) return 0;
  WBX_REQUIRE(acc != nullptr, "acc is NULL");
  WBX_HIP(hipSetDevice(ctx->device));
  WBX_HIP(hipMemsetAsync(acc, 0, (size_t)n * sizeof(double), ctx->stream));
  return 0;
}
