// comm.hip — the data-parallel collective of the train step on RCCL (xGMI): ONE in-place all-reduce (sum) of the P + 4 floats behind
// `grad` per optimiser step, enqueued on the step's own HIP stream between the partial-row reduction and the clip + optimiser kernel —
// so a whole epoch of sharded steps is issued from C++ without returning to the host (odpd_train_epoch_dp, capi.hip).
// The reference is single-device; partitioning = SURVEY §8(e): every rank holds a replica of the ~1k parameters and of
// the optimiser state, takes a contiguous shard of each global batch, normalises its loss gradient by the GLOBAL element count, and
// the sum over ranks is the global-batch gradient (uneven shards included); clip_grad_norm_ then sees the global norm.
// librccl is loaded with dlopen at the first communicator: single-GPU runs neither link nor initialise it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "odpd_host.h"

namespace odpd {
namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r = [] {
        Rccl v;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            v.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (v.handle) break;
        }
        if (!v.handle) return v;
        v.GetUniqueId = (decltype(v.GetUniqueId))dlsym(v.handle, "ncclGetUniqueId");
        v.CommInitRank = (decltype(v.CommInitRank))dlsym(v.handle, "ncclCommInitRank");
        v.CommDestroy = (decltype(v.CommDestroy))dlsym(v.handle, "ncclCommDestroy");
        v.AllReduce = (decltype(v.AllReduce))dlsym(v.handle, "ncclAllReduce");
        v.GetErrorString = (decltype(v.GetErrorString))dlsym(v.handle, "ncclGetErrorString");
        v.ok = v.GetUniqueId && v.CommInitRank && v.CommDestroy && v.AllReduce;
        return v;
    }();
    return r;
}
}  // namespace

struct Comm { ncclComm_t c; int rank, world; };

int comm_allreduce(hipStream_t st, void* comm, float* buf, int64_t n) {
    Comm* cm = static_cast<Comm*>(comm);
    if (!cm || !buf || n <= 0) return ODPD_EINVAL;
    if (cm->world == 1) return 0;                       // a sum over one rank
    const ncclResult_t r = rccl().AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, cm->c, st);
    return r == ncclSuccess ? 0 : ODPD_ECOMM;
}
int comm_rank(void* comm) { return static_cast<Comm*>(comm)->rank; }
int comm_world(void* comm) { return static_cast<Comm*>(comm)->world; }
}  // namespace odpd

using namespace odpd;

extern "C" int odpd_comm_unique_id(void* id128) {
    if (!id128) return ODPD_EINVAL;
    if (!rccl().ok) return ODPD_ECOMM;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return ODPD_ECOMM;
    static_assert(sizeof(id) == 128, "odpd_comm_unique_id hands out 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return 0;
}
extern "C" int odpd_comm_init(const void* id128, int world, int rank, void** comm_out) {
    if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) return ODPD_EINVAL;
    if (!rccl().ok) return ODPD_ECOMM;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    Comm* cm = new Comm{nullptr, rank, world};
    if (rccl().CommInitRank(&cm->c, world, id, rank) != ncclSuccess) { delete cm; return ODPD_ECOMM; }
    *comm_out = cm;
    return 0;
}
extern "C" int odpd_comm_destroy(void* comm) {
    Comm* cm = static_cast<Comm*>(comm);
    if (!cm) return ODPD_EINVAL;
    const ncclResult_t r = rccl().CommDestroy(cm->c);
    delete cm;
    return r == ncclSuccess ? 0 : ODPD_ECOMM;
}
extern "C" int odpd_comm_allreduce_sum(void* stream, void* comm, float* buf, int64_t n) {
    if (!comm) return ODPD_EINVAL;
    if (static_cast<Comm*>(comm)->world > 1 && !rccl().ok) return ODPD_ECOMM;
    return comm_allreduce((hipStream_t)stream, comm, buf, n);
}
