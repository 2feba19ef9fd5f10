// comm.hip — the data-parallel collective of the train step: ONE in-place sum of the P + 4 floats behind `grad` per optimiser step,
// enqueued on the step's own HIP stream between the partial-row reduction and the clip + optimiser kernel — so a whole epoch of
// sharded steps is issued from C++ without returning to the host (odpd_train_epoch_dp, capi.hip).
// The reference is single-device; partitioning = SURVEY §8(e): every rank holds a replica of the ~1k parameters and of
// the optimiser state, takes a contiguous shard of each global batch, normalises its loss gradient by the GLOBAL element count, and
// the sum over ranks is the global-batch gradient (uneven shards included); clip_grad_norm_ then sees the global norm.
//
// Two transports behind one handle:
//   * the one-shot exchange (odpd_xchg.h): every rank writes its vector into the peers' slots — peer HBM mapped with hipIpc (xGMI) or
//     a host shared-memory segment — and sums what arrived in its own; folded into the optimiser kernel's prologue (optim.hip), a
//     stand-alone one-workgroup kernel otherwise.  SURVEY §5's choice for this 4 KB message.
//   * RCCL's all-reduce (librccl loaded with dlopen at the first communicator: single-GPU runs neither link nor initialise it; its
//     few types are declared here so that the build does not need the RCCL development headers).
#include <dlfcn.h>
#include <fcntl.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <unistd.h>

#include "odpd_host.h"
#include "odpd_xchg.h"

namespace odpd {
namespace {
// ---- the slice of the RCCL ABI this file calls (rccl.h: ncclUniqueId = 128 opaque bytes, ncclFloat32 = 7, ncclSum = 0, ncclSuccess = 0)
struct RcclUniqueId { char internal[128]; };
typedef void* RcclComm;
constexpr int kRcclSuccess = 0, kRcclFloat32 = 7, kRcclSum = 0;
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
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
        v.ok = v.GetUniqueId && v.CommInitRank && v.CommDestroy && v.AllReduce;
        return v;
    }();
    return r;
}

enum { KIND_RCCL = 0, KIND_XCHG_IPC = 1, KIND_XCHG_SHM = 2 };
struct Comm {
    int rank = 0, world = 1, kind = KIND_RCCL;
    RcclComm c = nullptr;
    // one-shot exchange
    unsigned seq = 0;                              // advanced per exchange, the same sequence on every rank
    unsigned prev_seq = 0;                         // ... its value before the last advance (comm_xchg_rollback)
    unsigned long long* slots[kXchgMaxWorld] = {};  // every rank's slot base as mapped in this process (slots[rank] = own)
    bool opened[kXchgMaxWorld] = {};               // hipIpcOpenMemHandle mappings to close
    int* err = nullptr;
    long long timeout_ticks = 0;
    void* shm_host = nullptr;                      // KIND_XCHG_SHM: the mapping of the whole segment
    size_t shm_bytes = 0;
    char shm_name[96] = {};
    bool connected = false;
};
inline size_t slot_words(int world) { return (size_t)2 * world * kXchgMaxFloats; }

__global__ __launch_bounds__(1024) void xchg_allreduce_kernel(XchgDev xd, float* __restrict__ g, int n) { xchg_allreduce_block(xd, g, n); }
}  // namespace

int comm_rank(void* comm) { return static_cast<Comm*>(comm)->rank; }
int comm_world(void* comm) { return static_cast<Comm*>(comm)->world; }

// the next exchange of a one-shot communicator as a kernel argument (false: RCCL communicator, vector too long, not connected)
bool comm_next_xchg(void* comm, int64_t n, XchgDev* out) {
    Comm* cm = static_cast<Comm*>(comm);
    if (!cm || cm->kind == KIND_RCCL || !cm->connected || n <= 0 || n > kXchgMaxFloats) return false;
    // 0 is the state of a fresh slot.  The wrap goes to 2, not 1: 0xFFFFFFFF is odd, and two consecutive exchanges on the same parity would
    // break the two-parity overwrite argument of odpd_xchg.h
    cm->prev_seq = cm->seq;
    if (++cm->seq == 0) cm->seq = 2;
    const unsigned par = cm->seq & 1u;
    XchgDev xd{};
    for (int r = 0; r < cm->world; ++r) xd.dst[r] = cm->slots[r] + ((size_t)par * cm->world + cm->rank) * kXchgMaxFloats;
    xd.src = cm->slots[cm->rank] + (size_t)par * cm->world * kXchgMaxFloats;
    xd.err = cm->err;
    xd.timeout_ticks = cm->timeout_ticks;
    xd.seq = cm->seq;
    xd.world = cm->world; xd.rank = cm->rank; xd.row_stride = kXchgMaxFloats;
    *out = xd;
    return true;
}

// the exchange handed out last was never launched (the launch itself failed): give its sequence number back, so that this rank does not
// run one exchange ahead of its peers (they would spin until the time-out)
void comm_xchg_rollback(void* comm) {
    Comm* cm = static_cast<Comm*>(comm);
    if (!cm || cm->kind == KIND_RCCL) return;
    cm->seq = cm->prev_seq;
}

int comm_allreduce(hipStream_t st, void* comm, float* buf, int64_t n) {
    Comm* cm = static_cast<Comm*>(comm);
    if (!cm || !buf || n <= 0) return ODPD_EINVAL;
    if (cm->kind != KIND_RCCL) {
        if (!cm->connected) return ODPD_ECOMM;
        for (int64_t o = 0; o < n; o += kXchgMaxFloats) {      // (the step's vector is one piece; longer buffers go in pieces)
            const int64_t len = n - o < kXchgMaxFloats ? n - o : kXchgMaxFloats;
            XchgDev xd;
            if (!comm_next_xchg(comm, len, &xd)) return ODPD_ECOMM;
            hipLaunchKernelGGL(xchg_allreduce_kernel, dim3(1), dim3(1024), 0, st, xd, buf + o, (int)len);
            if (hipError_t e = hipGetLastError()) { comm_xchg_rollback(comm); return (int)e; }
        }
        return 0;
    }
    // (a world of one rank goes through ncclAllReduce as well: the one-GPU tests exercise the very call an 8-GPU job makes)
    return rccl().AllReduce(buf, buf, (size_t)n, kRcclFloat32, kRcclSum, cm->c, st) == kRcclSuccess ? 0 : ODPD_ECOMM;
}
}  // namespace odpd

using namespace odpd;

extern "C" int odpd_comm_unique_id(void* id128) {
    if (!id128) return ODPD_EINVAL;
    if (!rccl().ok) return ODPD_ECOMM;
    RcclUniqueId id;
    if (rccl().GetUniqueId(&id) != kRcclSuccess) return ODPD_ECOMM;
    memcpy(id128, &id, sizeof(id));
    return 0;
}
extern "C" int odpd_comm_init(const void* id128, int world, int rank, void** comm_out) {
    if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) return ODPD_EINVAL;
    if (!rccl().ok) return ODPD_ECOMM;
    RcclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    Comm* cm = new Comm;
    cm->rank = rank; cm->world = world; cm->kind = KIND_RCCL;
    if (rccl().CommInitRank(&cm->c, world, id, rank) != kRcclSuccess) { delete cm; return ODPD_ECOMM; }
    cm->connected = true;
    *comm_out = cm;
    return 0;
}

// ---- one-shot exchange: create (local slots) -> hand the 64-byte handles round (any host channel) -> connect ------------------------
static long long xchg_timeout_ticks() {
    // default 10 minutes — what torch.distributed's NCCL watchdog allows a collective: a rank that is merely late (rank 0 writing the
    // epoch's checkpoint and logs) must not poison the step; a rank that is gone ends the wait, and the run, in NaN losses
    const char* e = getenv("ODPD_XCHG_TIMEOUT_MS");
    const long long ms = e ? atoll(e) : 600000;
    return (ms > 0 ? ms : 600000) * 100000LL;      // wall_clock64() counts at 100 MHz
}
extern "C" int odpd_xchg_create(int world, int rank, const char* shm_name, void** comm_out, void* handle64_out) {
    if (!comm_out || world < 1 || world > kXchgMaxWorld || rank < 0 || rank >= world) return ODPD_EINVAL;
    if (!shm_name && !handle64_out) return ODPD_EINVAL;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "odpd_xchg_create hands out 64 bytes");
    Comm* cm = new Comm;
    cm->rank = rank; cm->world = world; cm->kind = shm_name ? KIND_XCHG_SHM : KIND_XCHG_IPC;
    cm->timeout_ticks = xchg_timeout_ticks();
    const size_t bytes = slot_words(world) * sizeof(unsigned long long);
    auto fail = [&](int rc) { odpd_comm_destroy(cm); return rc; };
    if (hipMalloc((void**)&cm->err, sizeof(int)) != hipSuccess) { cm->err = nullptr; return fail(ODPD_ECOMM); }
    if (hipMemset(cm->err, 0, sizeof(int)) != hipSuccess) return fail(ODPD_ECOMM);
    if (shm_name) {
        // one segment for the node: rank r's slots at r * bytes; every rank creates-or-opens it (a fresh segment reads as zeros)
        if (strlen(shm_name) >= sizeof(cm->shm_name)) return fail(ODPD_EINVAL);
        strcpy(cm->shm_name, shm_name);
        const int fd = shm_open(shm_name, O_CREAT | O_RDWR, 0600);
        if (fd < 0) return fail(ODPD_ECOMM);
        cm->shm_bytes = bytes * world;
        if (ftruncate(fd, (off_t)cm->shm_bytes) != 0) { close(fd); return fail(ODPD_ECOMM); }
        void* p = mmap(nullptr, cm->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return fail(ODPD_ECOMM);
        cm->shm_host = p;
        if (hipHostRegister(p, cm->shm_bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) {
            munmap(p, cm->shm_bytes); cm->shm_host = nullptr;
            return fail(ODPD_ECOMM);
        }
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) return fail(ODPD_ECOMM);
        for (int r = 0; r < world; ++r) cm->slots[r] = (unsigned long long*)((char*)d + (size_t)r * bytes);
    } else {
        // uncached device memory: coherent for the peers' system-scope stores and this rank's polls while kernels run
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) return fail(ODPD_ECOMM);
        cm->slots[rank] = (unsigned long long*)p;
        if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(ODPD_ECOMM);
        if (world > 1) {
            hipIpcMemHandle_t h;
            if (hipIpcGetMemHandle(&h, p) != hipSuccess) return fail(ODPD_ECOMM);
            memcpy(handle64_out, &h, sizeof(h));
        } else {
            memset(handle64_out, 0, 64);
        }
    }
    *comm_out = cm;
    return 0;
}
// handles: world x 64 bytes in rank order (ignored for the shared-memory transport).  Call after EVERY rank's odpd_xchg_create returned.
extern "C" int odpd_xchg_connect(void* comm, const void* handles) {
    Comm* cm = static_cast<Comm*>(comm);
    if (!cm || cm->kind == KIND_RCCL) return ODPD_EINVAL;
    if (cm->kind == KIND_XCHG_IPC) {
        if (!handles && cm->world > 1) return ODPD_EINVAL;
        for (int r = 0; r < cm->world; ++r) {
            if (r == cm->rank || cm->opened[r]) continue;
            hipIpcMemHandle_t h;
            memcpy(&h, (const char*)handles + (size_t)r * 64, sizeof(h));
            void* p = nullptr;
            if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return ODPD_ECOMM; }
            cm->slots[r] = (unsigned long long*)p;
            cm->opened[r] = true;
        }
    }
    cm->connected = true;
    return 0;
}
// every rank is mapped (the caller's barrier says so): the segment's name can go, the mappings keep it alive
extern "C" int odpd_xchg_unlink(void* comm) {
    Comm* cm = static_cast<Comm*>(comm);
    if (!cm) return ODPD_EINVAL;
    if (cm->kind == KIND_XCHG_SHM && cm->shm_name[0]) { shm_unlink(cm->shm_name); cm->shm_name[0] = 0; }
    return 0;
}
extern "C" int odpd_comm_destroy(void* comm) {
    Comm* cm = static_cast<Comm*>(comm);
    if (!cm) return ODPD_EINVAL;
    int rc = 0;
    if (cm->kind == KIND_RCCL) {
        if (cm->c && rccl().CommDestroy(cm->c) != kRcclSuccess) rc = ODPD_ECOMM;
    } else {
        (void)hipDeviceSynchronize();
        if (cm->kind == KIND_XCHG_IPC) {
            for (int r = 0; r < cm->world; ++r)
                if (cm->opened[r]) (void)hipIpcCloseMemHandle(cm->slots[r]);
            if (cm->slots[cm->rank]) (void)hipFree(cm->slots[cm->rank]);
        } else if (cm->shm_host) {
            (void)hipHostUnregister(cm->shm_host);
            munmap(cm->shm_host, cm->shm_bytes);
            if (cm->shm_name[0] && cm->rank == 0) shm_unlink(cm->shm_name);
        }
        if (cm->err) (void)hipFree(cm->err);
    }
    delete cm;
    return rc;
}
// time-out of the one-shot exchange's wait for a peer (ms <= 0: back to $ODPD_XCHG_TIMEOUT_MS / the default); no-op for RCCL communicators
extern "C" int odpd_comm_set_timeout_ms(void* comm, int64_t ms) {
    Comm* cm = static_cast<Comm*>(comm);
    if (!cm) return ODPD_EINVAL;
    cm->timeout_ticks = ms > 0 ? ms * 100000LL : xchg_timeout_ticks();
    return 0;
}
extern "C" int odpd_comm_kind(void* comm) { return comm ? static_cast<Comm*>(comm)->kind : ODPD_EINVAL; }
// exchanges of this rank in which a peer's row did not arrive within the time-out (their sums were poisoned with NaN); synchronises
extern "C" int odpd_comm_errors(void* comm) {
    Comm* cm = static_cast<Comm*>(comm);
    if (!cm) return ODPD_EINVAL;
    if (cm->kind == KIND_RCCL || !cm->err) return 0;
    int v = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&v, cm->err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return ODPD_ECOMM;
    return v;
}
extern "C" int odpd_comm_allreduce_sum(void* stream, void* comm, float* buf, int64_t n) {
    if (!comm) return ODPD_EINVAL;
    return comm_allreduce((hipStream_t)stream, comm, buf, n);
}
