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


def init(backend=None, device=None, single=False):
    """Initialise the default process group when launched by torchrun; no-op for a single process — unless single=True and the
    process runs under torchrun (RANK set): a group of one rank then goes through the same rendezvous and backend as a real job."""
    import torch.distributed as dist
    rank, local, world = env_world()
    if (world > 1 or (single and "RANK" in os.environ)) and not dist.is_initialized():
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


def _group_device(device):
    """where control-plane tensors of the default group live: the GPU for an RCCL ("nccl") group, the host for gloo"""
    import torch.distributed as dist
    return device if dist.get_backend() == "nccl" else torch.device("cpu")


def _agree(ok, device):
    """True on every rank iff `ok` on every rank (MIN over the default group; trivially `ok` for one process).  Every step of building
    a communicator that can fail on ONE rank is followed by this, so that all ranks take the same branch — a rank that fell back to
    another collective on its own would meet its peers in mismatched collectives (a hang or a corrupted gradient)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_group_device(device))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


class NativeComm:
    """Communicator owned by libopendpd_hip.so (csrc/comm.hip): the step's one all-reduce is enqueued from C++ on the step's own
    stream — folded into the optimiser kernel (`odpd_clip_optim_step_dp`), per buffer (`odpd_comm_allreduce_sum`), or for a whole sharded
    epoch (`odpd_train_epoch_dp` / `odpd_train_epoch_cascade`) — instead of going through torch.distributed from Python.  `kind`:
      "xchg"     one-shot exchange through peer HBM mapped with hipIpc (xGMI; also ranks sharing one GPU)
      "xchg_shm" one-shot exchange through a host shared-memory segment
      "rccl"     RCCL's all-reduce
    Construction is COLLECTIVE and never raises on one rank alone: every fallible stage ends in `_agree`, `self.ok` is the common
    verdict, and a failed communicator is torn down on all ranks together.  The default process group only carries the 64/128-byte
    handles and the verdicts (the only use torch.distributed has on this path)."""

    def __init__(self, device, kind):
        self._build(device, kind)
        candidate_log.append({"kind": kind, "ok": bool(self.ok), "why": self.why})

    def _build(self, device, kind):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib
        lib = _lib.load()
        self._lib, self.kind, self.device, self.handle, self.ok, self.why = lib, kind, device, None, False, ""
        self.rank, _, self.world = env_world()
        multi = dist.is_available() and dist.is_initialized()
        if multi:
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
        multi = multi and self.world > 1
        cdev = _group_device(device) if multi else torch.device("cpu")
        handle = C.c_void_p()
        import contextlib
        with (torch.cuda.device(device) if torch.device(device).type == "cuda" else contextlib.nullcontext()):
            if kind == "rccl":
                # rank 0 draws the id and ALWAYS broadcasts (zeros when it failed); nobody enters ncclCommInitRank unless rank 0 succeeded
                buf = (C.c_ubyte * 128)()
                ok0 = self.rank != 0 or lib.odpd_comm_unique_id(C.cast(buf, C.c_void_p)) == 0
                ident = torch.tensor(list(buf), dtype=torch.uint8)
                if multi:
                    carrier = ident.to(cdev)
                    dist.broadcast(carrier, src=0)
                    ident = carrier.cpu()
                if not _agree(ok0, device):
                    self.why = "ncclGetUniqueId failed on rank 0 (librccl missing?)"
                    return
                raw = (C.c_ubyte * 128)(*ident.tolist())
                rc = lib.odpd_comm_init(C.cast(raw, C.c_void_p), self.world, self.rank, C.byref(handle))
                self.handle = handle if rc == 0 else None
                if not _agree(rc == 0, device):
                    self.why = "ncclCommInitRank failed on a rank"
                    self.close(sync=False)
                    return
            else:
                name = None
                if kind == "xchg_shm":      # one segment name for the node, drawn by rank 0
                    tag = torch.tensor([os.getpid(), NativeComm._count], dtype=torch.int64)
                    if multi:
                        tag = tag.to(cdev)
                        dist.broadcast(tag, src=0)
                    name = f"/odpd_xchg_{int(tag[0])}_{int(tag[1])}".encode()
                NativeComm._count += 1
                h64 = (C.c_ubyte * 64)()
                rc = lib.odpd_xchg_create(self.world, self.rank, name, C.byref(handle), C.cast(h64, C.c_void_p))
                self.handle = handle if rc == 0 else None
                mine = torch.tensor(list(h64), dtype=torch.uint8)
                if multi:       # every rank takes part in the gather whatever its own outcome was
                    rows = [torch.zeros(64, dtype=torch.uint8, device=cdev) for _ in range(self.world)]
                    dist.all_gather(rows, mine.to(cdev))
                    allh = torch.cat([r.cpu() for r in rows])
                else:
                    allh = mine
                if not _agree(rc == 0, device):
                    self.why = "odpd_xchg_create failed on a rank (slot allocation / hipIpcGetMemHandle / shared-memory segment)"
                    self.close(sync=False)
                    return
                raw = (C.c_ubyte * (64 * self.world))(*allh.tolist())
                rc = lib.odpd_xchg_connect(self.handle, C.cast(raw, C.c_void_p))
                if not _agree(rc == 0, device):
                    self.why = "odpd_xchg_connect failed on a rank (hipIpcOpenMemHandle: no peer access?)"
                    self.close(sync=False)
                    return
                lib.odpd_xchg_unlink(self.handle)
            # self-test before the communicator is trusted with a gradient: three sums (both slot parities) of known vectors
            # (the ranks have just met in _agree: a peer that does not show up within seconds here never will)
            good = True
            if kind != "rccl":
                lib.odpd_comm_set_timeout_ms(self.handle, int(os.environ.get("ODPD_XCHG_SELFTEST_TIMEOUT_MS", "10000")))
            try:
                n = 1045
                base = torch.arange(n, dtype=torch.float32, device=device)
                for k in range(3):
                    buf = base * (self.rank + 1) + k
                    self.allreduce_sum_(buf)
                    want = base * (self.world * (self.world + 1) // 2) + k * self.world
                    good = good and bool(torch.equal(buf, want))
                good = good and lib.odpd_comm_errors(self.handle) == 0
            except Exception as exc:      # noqa: BLE001 — any failure here is a verdict, not a crash of one rank
                good, self.why = False, f"self-test raised {exc!r}"
            if not _agree(good, device):
                self.why = self.why or "self-test all-reduce returned a wrong sum / timed out on a rank"
                self.close(sync=False)
                return
            if kind != "rccl":
                lib.odpd_comm_set_timeout_ms(self.handle, 0)       # from here on: $ODPD_XCHG_TIMEOUT_MS / ten minutes
        self.ok = True

    _count = 0

    def allreduce_sum_(self, t):
        from . import _lib
        _lib.check(self._lib.odpd_comm_allreduce_sum(_lib.stream_ptr(), self.handle, _lib.ptr(t), t.numel()), "odpd_comm_allreduce_sum")
        return t

    def errors(self):
        """exchanges of this rank that timed out (their sums are NaN); synchronises the device"""
        return int(self._lib.odpd_comm_errors(self.handle)) if self.handle else 0

    def describe(self):
        return {"xchg": "one-shot exchange: every rank writes its P+4 floats into the peers' hipIpc-mapped HBM slots (xGMI) and sums its own "
                        "slots in rank order, inside the optimiser kernel (csrc/odpd_xchg.h)",
                "xchg_shm": "one-shot exchange through a host shared-memory segment, inside the optimiser kernel (csrc/odpd_xchg.h)",
                "rccl": "RCCL all-reduce of P+4 floats per step, enqueued by libopendpd_hip.so on the step's stream (csrc/comm.hip)"}[self.kind]

    def close(self, sync=True):
        """collective when sync=True: no rank frees the slots its peers may still be writing to"""
        import torch.distributed as dist
        if self.handle:
            if sync:
                if torch.device(self.device).type == "cuda":
                    torch.cuda.synchronize(self.device)
                if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    dist.barrier()
            self._lib.odpd_comm_destroy(self.handle)
            self.handle = None
        self.ok = False


_native = None
candidate_log = []      # every NativeComm this process built: {"kind", "ok", "why"} (bench.py prints it: which candidate failed, and where)


def comm_candidates(backend, multi):
    """Which library-owned communicators to try, in order, from $ODPD_COMM (auto | xchg | xchg_shm | rccl | torch) and the legacy
    $ODPD_NATIVE_COMM (0 = torch.distributed; 1 = also for a single process, which exercises the path on a one-GPU box).
    auto: on an RCCL ("nccl") default group — one process per GPU — the one-shot exchange, then RCCL; on a gloo group nothing (ranks
    may share a device there, which RCCL refuses: such groups ask for `xchg` / `xchg_shm` explicitly).  A single process only when asked."""
    mode = os.environ.get("ODPD_COMM", "auto").lower()
    legacy = os.environ.get("ODPD_NATIVE_COMM")
    if mode not in ("auto", "xchg", "xchg_shm", "rccl", "torch"):
        raise ValueError(f"ODPD_COMM={mode!r}: expected auto, xchg, xchg_shm, rccl or torch")
    if legacy == "0" or mode == "torch":
        return []
    if mode != "auto":
        return [mode]
    if multi:
        return ["xchg", "rccl"] if backend == "nccl" else []
    return ["rccl"] if legacy == "1" else []


def native_comm(device=None):
    """The process-wide NativeComm, created collectively on first use (see comm_candidates); None when no library-owned communicator
    is wanted or none passed its self-test on EVERY rank — the caller then uses torch.distributed."""
    global _native
    import torch.distributed as dist
    if _native is not None:
        return _native or None
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    kinds = comm_candidates(dist.get_backend() if multi else None, multi)
    if not kinds:
        return None
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    _native = False
    for kind in kinds:
        comm = NativeComm(device, kind)
        if comm.ok:
            _native = comm
            break
        if comm.rank == 0:
            print(f"[opendpd_amd] {kind} communicator unavailable ({comm.why}); trying the next collective")
    if _native is False and (not multi or dist.get_rank() == 0):
        print("[opendpd_amd] no library-owned communicator; using torch.distributed for the gradient all-reduce")
    return _native or None


def reset_native_comm():
    """drop the process-wide communicator (collective); the next native_comm() call builds a new one"""
    global _native
    if _native:
        _native.close()
    _native = None
