// RCCL exchanges of the frame-axis shard behind the C ABI (`allgather_frames`; 8e: cfg 4, one long clip over the
// GPUs of a node).  The all-gather north_star names for temporal attention, and the all-to-all (frame shard <-> pixel shard)
// the engine prefers -- on the xGMI mesh every pair of GPUs has its own link, so a rank's seven transfers run concurrently --
// on a communicator the library owns: one process per GPU, the 128-byte unique id travels through whatever the host already
// has (torch.distributed's store, MPI, a file).  RCCL is resolved at run time (the copy already loaded into the process --
// PyTorch-ROCm ships one -- else librccl.so.1): the compute kernels never depend on it and a box without RCCL still loads
// the library.  Calls are asynchronous on the passed stream and capturable into a hipGraph like every other entry.
#include <dlfcn.h>

#include "common.h"

namespace {

typedef struct { char internal[128]; } rccl_unique_id;  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_get_id)(rccl_unique_id*);
typedef int (*fn_init_rank)(void**, int, rccl_unique_id, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_alltoall)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*fn_errstr)(int);

struct Rccl {
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_destroy destroy = nullptr;
  fn_allgather allgather = nullptr;
  fn_alltoall alltoall = nullptr;
  fn_errstr errstr = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclAllGather")) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) return x;
    }
    x.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
    x.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    x.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    x.allgather = (fn_allgather)dlsym(h, "ncclAllGather");
    x.alltoall = (fn_alltoall)dlsym(h, "ncclAllToAll");
    x.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    x.ok = x.get_id && x.init_rank && x.destroy && x.allgather && x.alltoall;
    return x;
  }();
  return r;
}

constexpr int kNcclChar = 0;  // ncclInt8 / ncclChar

int fail(const char* what, int rc) {
  const Rccl& r = rccl();
  mvoc_set_error("%s: RCCL error %d (%s)", what, rc, r.errstr ? r.errstr(rc) : "?");
  return -3;
}

}  // namespace

extern "C" int mvoc_comm_unique_id(void* id128) {
  MVOC_REQUIRE(id128, -1, "comm_unique_id: null");
  MVOC_REQUIRE(rccl().ok, -3, "RCCL is not available in this process (librccl.so.1 not found)");
  rccl_unique_id id;
  if (int rc = rccl().get_id(&id)) return fail("ncclGetUniqueId", rc);
  memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int mvoc_comm_init(const void* id128, int32_t rank, int32_t world, void** comm) {
  MVOC_REQUIRE(id128 && comm && world >= 1 && rank >= 0 && rank < world, -1, "comm_init: bad args");
  MVOC_REQUIRE(rccl().ok, -3, "RCCL is not available in this process (librccl.so.1 not found)");
  rccl_unique_id id;
  memcpy(&id, id128, sizeof(id));
  if (int rc = rccl().init_rank(comm, world, id, rank)) return fail("ncclCommInitRank", rc);
  return 0;
}

extern "C" int mvoc_comm_destroy(void* comm) {
  MVOC_REQUIRE(comm && rccl().ok, -1, "comm_destroy: bad args");
  if (int rc = rccl().destroy(comm)) return fail("ncclCommDestroy", rc);
  return 0;
}

extern "C" int mvoc_allgather_frames(void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
  MVOC_REQUIRE(comm && send && recv && bytes_per_rank > 0 && rccl().ok, -1, "allgather_frames: bad args");
  if (int rc = rccl().allgather(send, recv, bytes_per_rank, kNcclChar, comm, (hipStream_t)stream)) return fail("ncclAllGather", rc);
  return 0;
}

extern "C" int mvoc_alltoall_frames(void* comm, const void* send, void* recv, size_t bytes_per_peer, void* stream) {
  MVOC_REQUIRE(comm && send && recv && bytes_per_peer > 0 && rccl().ok, -1, "alltoall_frames: bad args");
  if (int rc = rccl().alltoall(send, recv, bytes_per_peer, kNcclChar, comm, (hipStream_t)stream)) return fail("ncclAllToAll", rc);
  return 0;
}
