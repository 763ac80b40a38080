// Data parallelism through the C ABI: a communicator of the library's own over RCCL and the in-place sum of a flat fp32 gradient buffer
// over the ranks, enqueued on the caller's stream -- what the reference gets from torch.nn.DataParallel's gradient reduction
// (models/model_util.py:283-284) and SURVEY.md section 8(b) names mcdseg_allreduce(buf, count, comm, stream).
//
// RCCL is bound at run time, not at link time: a process that trains with PyTorch has already loaded a librccl (PyTorch ships its
// own), and a second copy of the library in one address space is the one thing to avoid -- so the entry points look for the loaded one
// first (dlopen RTLD_NOLOAD), then for librccl.so.1 on the loader's path, and report -ENOSYS when there is none.  Nothing here touches
// the GPU before the caller asks for a communicator.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace {

struct McdNcclId {
  char internal[128];  // NCCL_UNIQUE_ID_BYTES (rccl.h)
};
static_assert(sizeof(McdNcclId) == MCDSEG_COMM_ID_BYTES, "id size");

// the five RCCL entry points the path needs (rccl.h; ncclResult_t is an int-sized enum, ncclSuccess = 0)
typedef int (*get_unique_id_fn)(McdNcclId*);
typedef int (*comm_init_rank_fn)(void**, int, McdNcclId, int);
typedef int (*comm_destroy_fn)(void*);
typedef int (*all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*error_string_fn)(int);

struct Rccl {
  void* handle = nullptr;
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  all_reduce_fn all_reduce = nullptr;
  error_string_fn error_string = nullptr;
  bool ok = false;
};

const Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (int pass = 0; pass < 2 && r.handle == nullptr; ++pass)  // pass 0: only a copy the process has loaded already
      for (const char* n : names)
        if (r.handle == nullptr) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
    if (r.handle == nullptr) return;
    r.get_unique_id = (get_unique_id_fn)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (comm_init_rank_fn)dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (comm_destroy_fn)dlsym(r.handle, "ncclCommDestroy");
    r.all_reduce = (all_reduce_fn)dlsym(r.handle, "ncclAllReduce");
    r.error_string = (error_string_fn)dlsym(r.handle, "ncclGetErrorString");
    r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce;
  });
  return r;
}

int need_rccl(const char* who) {
  if (rccl().ok) return 0;
  mcdseg_set_error("%s: no RCCL in this process (librccl.so.1 not loaded and not on the loader's path)", who);
  return -38;  // -ENOSYS
}

int fail(const char* who, int rc) {
  const Rccl& r = rccl();
  mcdseg_set_error("%s: RCCL error %d (%s)", who, rc, r.error_string ? r.error_string(rc) : "?");
  return -5;  // -EIO
}

}  // namespace

extern "C" int mcdseg_comm_unique_id(void* id128) {
  MCD_REQUIRE(id128 != nullptr, "mcdseg_comm_unique_id: null id buffer");
  if (int rc = need_rccl("mcdseg_comm_unique_id")) return rc;
  McdNcclId id;
  if (int rc = rccl().get_unique_id(&id)) return fail("mcdseg_comm_unique_id", rc);
  memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int mcdseg_comm_init(void** comm, int32_t nranks, const void* id128, int32_t rank) {
  MCD_REQUIRE(comm != nullptr && id128 != nullptr, "mcdseg_comm_init: null argument");
  MCD_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "mcdseg_comm_init: rank %d of %d", rank, nranks);
  if (int rc = need_rccl("mcdseg_comm_init")) return rc;
  McdNcclId id;
  memcpy(&id, id128, sizeof(id));
  *comm = nullptr;
  if (int rc = rccl().comm_init_rank(comm, nranks, id, rank)) return fail("mcdseg_comm_init", rc);
  return 0;
}

extern "C" int mcdseg_comm_destroy(void* comm) {
  if (comm == nullptr) return 0;
  if (int rc = need_rccl("mcdseg_comm_destroy")) return rc;
  if (int rc = rccl().comm_destroy(comm)) return fail("mcdseg_comm_destroy", rc);
  return 0;
}

extern "C" int mcdseg_allreduce(float* buf, int64_t count, void* comm, void* stream) {
  MCD_REQUIRE(comm != nullptr, "mcdseg_allreduce: null communicator (mcdseg_comm_init first)");
  MCD_REQUIRE(count >= 0 && (buf != nullptr || count == 0), "mcdseg_allreduce: null buffer");
  if (count == 0) return 0;
  if (int rc = need_rccl("mcdseg_allreduce")) return rc;
  // in place, fp32 (ncclFloat32 = 7), sum (ncclSum = 0); asynchronous on the caller's stream like every other entry point
  if (int rc = rccl().all_reduce(buf, buf, (size_t)count, 7, 0, comm, (hipStream_t)stream)) return fail("mcdseg_allreduce", rc);
  return 0;
}
