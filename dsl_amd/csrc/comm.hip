// The data-parallel exchange of the training step behind the C-ABI: an RCCL communicator per process (one process per
// GPU, ring / tree over xGMI chosen by RCCL) and in-place fp32 sum all-reduces of contiguous ranges of the flat gradient
// buffer.  Stands where the reference has MMDistributedDataParallel's bucket hooks (mmdet/apis/train.py:92-96) and
// reduce_mean (mmdet/core/utils/dist_utils.py:63-69).
//
// librccl is bound at run time (dlopen of its soname: inside a PyTorch process that is the copy torch already holds), so
// libdsl_hip.so itself loads on a box without RCCL and a single-GPU user never touches it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.hpp"

namespace {

struct Rccl {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  decltype(&ncclCommCount) comm_count = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;
char g_load_err[256] = "symbols missing";

const Rccl* rccl() {
  std::call_once(g_once, [] {
    Rccl& r = g_rccl;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.h) break;
      const char* e = dlerror();
      if (e) snprintf(g_load_err, sizeof(g_load_err), "%s", e);
    }
    if (!r.h) return;
    r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.h, "ncclGetUniqueId");
    r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.h, "ncclCommInitRank");
    r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.h, "ncclCommDestroy");
    r.all_reduce = (decltype(r.all_reduce))dlsym(r.h, "ncclAllReduce");
    r.group_start = (decltype(r.group_start))dlsym(r.h, "ncclGroupStart");
    r.group_end = (decltype(r.group_end))dlsym(r.h, "ncclGroupEnd");
    r.error_string = (decltype(r.error_string))dlsym(r.h, "ncclGetErrorString");
    r.comm_count = (decltype(r.comm_count))dlsym(r.h, "ncclCommCount");
    r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce && r.group_start && r.group_end &&
           r.error_string && r.comm_count;
  });
  return g_rccl.ok ? &g_rccl : nullptr;
}

#define RCCL_OR_FAIL(r)                                                                          \
  const Rccl* r = rccl();                                                                        \
  DSL_CHECK(r, "librccl.so.1 could not be bound (%s)", g_load_err)

#define RCCL_CALL(r, what, expr)                                           \
  do {                                                                     \
    const ncclResult_t rc_ = (expr);                                       \
    DSL_CHECK(rc_ == ncclSuccess, "%s: %s", what, (r)->error_string(rc_)); \
  } while (0)

}  // namespace

extern "C" int dsl_comm_unique_id(void* id128) {
  DSL_CHECK(id128, "dsl_comm_unique_id: null pointer");
  RCCL_OR_FAIL(r);
  static_assert(sizeof(ncclUniqueId) == 128, "the header promises a 128-byte id");
  ncclUniqueId id;
  RCCL_CALL(r, "ncclGetUniqueId", r->get_unique_id(&id));
  memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int dsl_comm_init_rank(void** comm, int nranks, const void* id128, int rank) {
  DSL_CHECK(comm && id128 && nranks >= 1 && rank >= 0 && rank < nranks, "dsl_comm_init_rank: bad arguments");
  RCCL_OR_FAIL(r);
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  RCCL_CALL(r, "ncclCommInitRank", r->comm_init_rank(&c, nranks, id, rank));
  *comm = (void*)c;
  return 0;
}

extern "C" int dsl_comm_size(void* comm) {
  DSL_CHECK(comm, "dsl_comm_size: null communicator");
  RCCL_OR_FAIL(r);
  int n = 0;
  RCCL_CALL(r, "ncclCommCount", r->comm_count((ncclComm_t)comm, &n));
  return n;
}

extern "C" int dsl_comm_destroy(void* comm) {
  if (!comm) return 0;
  RCCL_OR_FAIL(r);
  RCCL_CALL(r, "ncclCommDestroy", r->comm_destroy((ncclComm_t)comm));
  return 0;
}

extern "C" int dsl_allreduce_bucket(void* comm, float* buf, size_t count, void* stream) {
  DSL_CHECK(comm && (buf || count == 0), "dsl_allreduce_bucket: null pointer");
  if (count == 0) return 0;
  RCCL_OR_FAIL(r);
  RCCL_CALL(r, "ncclAllReduce", r->all_reduce(buf, buf, count, ncclFloat, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
  return 0;
}

extern "C" int dsl_allreduce_bucket_bf16(void* comm, void* buf, size_t count, void* stream) {
  DSL_CHECK(comm && (buf || count == 0), "dsl_allreduce_bucket_bf16: null pointer");
  if (count == 0) return 0;
  RCCL_OR_FAIL(r);
  RCCL_CALL(r, "ncclAllReduce", r->all_reduce(buf, buf, count, ncclBfloat16, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
  return 0;
}

extern "C" int dsl_allreduce_buckets(void* comm, float* const* bufs, const size_t* counts, int n, void* stream) {
  DSL_CHECK(comm && bufs && counts && n >= 0, "dsl_allreduce_buckets: bad arguments");
  RCCL_OR_FAIL(r);
  RCCL_CALL(r, "ncclGroupStart", r->group_start());
  ncclResult_t rc = ncclSuccess;
  for (int i = 0; i < n && rc == ncclSuccess; ++i)
    if (counts[i]) rc = r->all_reduce(bufs[i], bufs[i], counts[i], ncclFloat, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
  const ncclResult_t rc2 = r->group_end();
  DSL_CHECK(rc == ncclSuccess, "ncclAllReduce: %s", r->error_string(rc));
  DSL_CHECK(rc2 == ncclSuccess, "ncclGroupEnd: %s", r->error_string(rc2));
  return 0;
}
