// comm.hip -- thin RCCL entry points of the multi-GPU layer (`mrag_allgather`): the ONE exchange step of the hot path
// is an all-gather -- the ranks' final latents at the end of the denoising loop (clip sharding), a rank's K / V rows per DiT block
// (sequence sharding), the velocity of a guidance branch (CFG pairs) -- issued on a caller-chosen HIP stream so it overlaps compute on
// another one (the reference reaches NCCL only through Lightning DDP: configs/cogvideox/MotionRAG_open.yml:4-8).
// RCCL is resolved at run time from the librccl.so the process already has (torch's), never linked: a single-GPU user of libmrag_hip.so
// needs no communication library, and two RCCL builds in one process are avoided.  No RCCL -> MRAG_ENOTSUP.
#include "../../include/mrag_hip.h"
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

namespace {

typedef struct { char internal[128]; } rccl_unique_id;           // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*get_unique_id_fn)(rccl_unique_id*);
typedef int (*comm_init_rank_fn)(void**, int, rccl_unique_id, int);
typedef int (*comm_destroy_fn)(void*);
typedef int (*all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);

struct Rccl {
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  all_gather_fn all_gather = nullptr;
  bool ok = false;
};

const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl x;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);     // the copy the process already loaded (torch.distributed's), if any
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return x;
    x.get_unique_id = (get_unique_id_fn)dlsym(h, "ncclGetUniqueId");
    x.comm_init_rank = (comm_init_rank_fn)dlsym(h, "ncclCommInitRank");
    x.comm_destroy = (comm_destroy_fn)dlsym(h, "ncclCommDestroy");
    x.all_gather = (all_gather_fn)dlsym(h, "ncclAllGather");
    x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.all_gather;
    return x;
  }();
  return r;
}

}  // namespace

extern "C" int mrag_comm_unique_id(void* id128_host) {
  if (!id128_host) return MRAG_EINVAL;
  if (!rccl().ok) return MRAG_ENOTSUP;
  rccl_unique_id id;
  const int rc = rccl().get_unique_id(&id);
  if (rc != 0) return 1000 + rc;
  memcpy(id128_host, id.internal, 128);
  return MRAG_OK;
}

extern "C" int mrag_comm_init(const void* id128_host, int32_t rank, int32_t world, void** comm_out) {
  if (!id128_host || !comm_out || world <= 0 || rank < 0 || rank >= world) return MRAG_EINVAL;
  if (!rccl().ok) return MRAG_ENOTSUP;
  rccl_unique_id id;
  memcpy(id.internal, id128_host, 128);
  const int rc = rccl().comm_init_rank(comm_out, world, id, rank);
  return rc == 0 ? MRAG_OK : 1000 + rc;
}

extern "C" int mrag_comm_destroy(void* comm) {
  if (!comm) return MRAG_EINVAL;
  if (!rccl().ok) return MRAG_ENOTSUP;
  const int rc = rccl().comm_destroy(comm);
  return rc == 0 ? MRAG_OK : 1000 + rc;
}

extern "C" int mrag_allgather(void* stream, void* comm, const void* send, void* recv, int64_t bytes_per_rank) {
  if (!comm || !send || !recv || bytes_per_rank <= 0) return MRAG_EINVAL;
  if (!rccl().ok) return MRAG_ENOTSUP;
  const int rc = rccl().all_gather(send, recv, (size_t)bytes_per_rank, /*ncclUint8*/ 1, comm, (hipStream_t)stream);
  return rc == 0 ? MRAG_OK : 1000 + rc;
}
