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
