"""Data-parallel plumbing: the batch of IQ frames is sharded over ranks (one process per GPU), every rank keeps a
full replica of the ~1k parameters and optimiser state, and ONE all-reduce (sum) of P+4 floats per step — the
gradient plus the loss partial sum — is the only collective.  The reference is single-device; the
contract that keeps it exact for uneven shards is: every rank normalises its loss gradient by the GLOBAL element
count, so the sum of rank gradients is the global-batch gradient; clip_grad_norm_ then sees the global norm.

`torch.distributed` backend "nccl" is RCCL on ROCm (xGMI); "gloo" is used by the CPU tests."""
import os

import torch


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process -> (0, 0, 1))."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device=None):
    """Initialise the default process group when launched by torchrun; no-op for a single process."""
    import torch.distributed as dist
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n items for `rank`; sizes differ by at most one (last batch 157 = 79 + 78)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(x, target, rank, world):
    """This rank's slice of a global batch and the global element count used for the loss mean."""
    lo, hi = shard_range(x.shape[0], rank, world)
    return x[lo:hi], target[lo:hi], x.shape[0] * x.shape[1] * x.shape[2]


def allreduce_sum_(t, group=None):
    """In-place sum over ranks (no-op for one process)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def broadcast_params_(module, src=0, group=None):
    """Make every replica start from rank `src`'s parameters."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for p in module.parameters():
            dist.broadcast(p.data, src=src, group=group)


class NativeComm:
    """RCCL communicator owned by libopendpd_hip.so (csrc/comm.hip): the step's one all-reduce is then enqueued from C++ on the step's
    own stream — `odpd_comm_allreduce_sum` per step, or the whole sharded epoch through `odpd_train_epoch_dp` — instead of going
    through torch.distributed from Python.  Created collectively: rank 0 draws the 128-byte id, the default process group carries it
    to the other ranks (the only use torch.distributed has on this path)."""

    def __init__(self, device):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib
        lib = _lib.load()
        self.rank, _, self.world = env_world()
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (C.c_ubyte * 128)()
            _lib.check(lib.odpd_comm_unique_id(C.cast(buf, C.c_void_p)), "odpd_comm_unique_id")
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        if self.world > 1:
            carrier = ident.to(device) if dist.get_backend() == "nccl" else ident
            dist.broadcast(carrier, src=0)
            ident = carrier.cpu()
        raw = (C.c_ubyte * 128)(*ident.tolist())
        handle = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.odpd_comm_init(C.cast(raw, C.c_void_p), self.world, self.rank, C.byref(handle)), "odpd_comm_init")
        self.handle, self._lib = handle, lib

    def allreduce_sum_(self, t):
        from . import _lib
        _lib.check(self._lib.odpd_comm_allreduce_sum(_lib.stream_ptr(), self.handle, _lib.ptr(t), t.numel()), "odpd_comm_allreduce_sum")
        return t

    def close(self):
        if self.handle:
            self._lib.odpd_comm_destroy(self.handle)
            self.handle = None


_native = None


def native_comm(device=None):
    """The process-wide NativeComm, created on first use when the default process group runs on RCCL ("nccl" backend: one process per
    GPU) — or for a single process when $ODPD_NATIVE_COMM=1 (exercises the RCCL path on a one-GPU box).  None otherwise (gloo groups:
    several ranks may share a device, which RCCL refuses) or when RCCL cannot be initialised: the caller then uses torch.distributed."""
    global _native
    import torch.distributed as dist
    if _native is not None:
        return _native or None
    want = os.environ.get("ODPD_NATIVE_COMM")
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if want == "0" or (not multi and want != "1") or (multi and dist.get_backend() != "nccl"):
        return None
    try:
        _native = NativeComm(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    except Exception as exc:      # missing librccl, refused topology: announce and fall back
        print(f"[opendpd_amd] native RCCL communicator unavailable ({exc}); using torch.distributed for the gradient all-reduce")
        _native = False
    return _native or None
