"""world_size-2 gloo test (CPU) of the data-parallel contract in opendpd_amd/dist.py: every rank computes the
gradient of ITS shard normalised by the GLOBAL element count, one all-reduce(sum) of P+4 floats reproduces the
full-batch gradient and loss, and the clip+AdamW step then leaves identical replicas.  The per-rank gradient comes
from the CPU oracle here (no GPU in this test); the collective and sharding code is the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from opendpd_amd import dist as odist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, T, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle.oracle import Oracle, make_model
    odist.init("gloo")
    o = Oracle("f32")
    m = make_model("dgru", 13)
    rng = np.random.RandomState(0)
    P = o.param_count(m)
    params = (rng.randn(P) * 0.2).astype(np.float32)
    x = (0.05 + 0.8 * rng.rand(B, T, 2)).astype(np.float32)
    t = rng.rand(B, T, 2).astype(np.float32)
    xs, ts, count = odist.shard_batch(torch.from_numpy(x), torch.from_numpy(t), rank, world)
    y, _ = o.forward(m, params, xs.numpy())
    loss_local, dy = o.loss("l2", y, ts.numpy(), count=count)     # normalised by the GLOBAL count
    g, _ = o.backward(m, params, xs.numpy(), dy, need_dx=False)
    buf = torch.zeros(P + 4)
    buf[:P] = torch.from_numpy(g)
    buf[P] = loss_local * count                                   # un-normalised partial sum (column P)
    odist.allreduce_sum_(buf)
    mom, var = np.zeros(P, np.float32), np.zeros(P, np.float32)
    gg = buf[:P].numpy().copy()
    o.clip_adamw(params, gg, mom, var, 1, 5e-4, 200.0)
    q.put((rank, buf.numpy().copy(), params.copy(), count))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])   # even and uneven shards
def test_two_rank_allreduce_matches_full_batch(B):
    from oracle.oracle import Oracle, make_model
    T, world, port = 20, 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    o = Oracle("f32")
    m = make_model("dgru", 13)
    rng = np.random.RandomState(0)
    P = o.param_count(m)
    params = (rng.randn(P) * 0.2).astype(np.float32)
    x = (0.05 + 0.8 * rng.rand(B, T, 2)).astype(np.float32)
    t = rng.rand(B, T, 2).astype(np.float32)
    y, _ = o.forward(m, params, x)
    loss, dy = o.loss("l2", y, t)
    g, _ = o.backward(m, params, x, dy, need_dx=False)
    for rank, buf, p_after, count in res:
        assert count == B * T * 2
        assert np.abs(buf[:P] - g).max() < 1e-6 * max(1.0, np.abs(g).max())
        assert abs(buf[P] / count - loss) < 1e-6
    assert np.array_equal(res[0][1], res[1][1])      # identical reduced buffers
    assert np.array_equal(res[0][2], res[1][2])      # identical replicas after the step


def test_shard_ranges_cover_batch():
    for n in (1, 7, 157, 256):
        for w in (1, 2, 4, 8):
            r = [odist.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


# ---- building the library-owned communicator is a COLLECTIVE decision (a rank that fell back on its own would meet its peers
# in mismatched collectives) ------------------------------------------------------------------------------------------------------------
class _FakeLib:
    """stands in for libopendpd_hip.so: every stage's return code is scripted per rank; records what was called"""

    def __init__(self, rank, script):
        self.rank, self.script, self.calls = rank, script, []

    def _rc(self, name):
        self.calls.append(name)
        return self.script.get(name, {}).get(self.rank, 0)

    def odpd_comm_unique_id(self, buf):
        return self._rc("unique_id")

    def odpd_comm_init(self, raw, world, rank, out):
        out._obj.value = 0x1000          # (a non-NULL handle, as the library hands out)
        return self._rc("init")

    def odpd_xchg_create(self, world, rank, name, out, h64):
        out._obj.value = 0x1000
        return self._rc("create")

    def odpd_xchg_connect(self, handle, raw):
        return self._rc("connect")

    def odpd_xchg_unlink(self, handle):
        return self._rc("unlink")

    def odpd_comm_destroy(self, handle):
        return self._rc("destroy")

    def odpd_comm_errors(self, handle):
        return self._rc("errors")

    def odpd_comm_allreduce_sum(self, *a):
        return self._rc("allreduce")

    def odpd_comm_set_timeout_ms(self, handle, ms):
        return 0


def _comm_worker(rank, world, port, kind, script, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ODPD_COMM=kind)
    from opendpd_amd import _lib
    odist.init("gloo")
    fake = _FakeLib(rank, script)
    _lib.load = lambda: fake
    _lib.stream_ptr = lambda: None
    _lib.ptr = lambda t: None
    comm = odist.native_comm(torch.device("cpu"))
    # whatever was decided, the ranks are still in step: a collective on the default group completes with the right sum
    buf = torch.full((5,), float(rank + 1))
    odist.allreduce_sum_(buf)
    q.put((rank, comm is not None, list(fake.calls), buf.tolist()))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("kind,script,stage_reached", [
    ("xchg", {"create": {1: -3}}, "create"),            # rank 1 cannot allocate / export its slots
    ("xchg", {"connect": {0: -3}}, "connect"),          # rank 0 cannot map a peer
    ("rccl", {"unique_id": {0: -3}}, "unique_id"),      # rank 0 has no librccl: nobody may enter ncclCommInitRank
    ("rccl", {"init": {1: -3}}, "init"),
    ("xchg_shm", {"errors": {1: 2}}, "errors"),         # the self-test saw time-outs on rank 1
])
def test_a_failure_on_one_rank_makes_every_rank_fall_back_together(kind, script, stage_reached):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, kind, script, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, has_comm, calls, buf in res:
        assert not has_comm, (rank, calls)                       # the common verdict: no library-owned communicator
        assert buf == [3.0] * 5                                   # and the default group still works (nobody is stuck in another collective)
        if stage_reached == "unique_id":
            assert "init" not in calls
        if stage_reached == "create":
            assert "connect" not in calls
    # a rank whose own stage succeeded tore its half down again
    ok_rank = [r for r in res if script[stage_reached].get(r[0], 0) == 0]
    if stage_reached in ("connect", "init", "errors"):
        assert all("destroy" in r[2] for r in res if stage_reached != "init" or r in ok_rank)


def test_comm_candidates_policy(monkeypatch):
    monkeypatch.delenv("ODPD_NATIVE_COMM", raising=False)
    monkeypatch.delenv("ODPD_COMM", raising=False)
    assert odist.comm_candidates("nccl", True) == ["xchg", "rccl"]
    assert odist.comm_candidates("gloo", True) == []
    assert odist.comm_candidates(None, False) == []
    monkeypatch.setenv("ODPD_NATIVE_COMM", "1")
    assert odist.comm_candidates(None, False) == ["rccl"]
    monkeypatch.setenv("ODPD_NATIVE_COMM", "0")
    assert odist.comm_candidates("nccl", True) == []
    monkeypatch.delenv("ODPD_NATIVE_COMM")
    for mode in ("xchg", "xchg_shm", "rccl"):
        monkeypatch.setenv("ODPD_COMM", mode)
        assert odist.comm_candidates("gloo", True) == [mode] and odist.comm_candidates(None, False) == [mode]
    monkeypatch.setenv("ODPD_COMM", "torch")
    assert odist.comm_candidates("nccl", True) == []
    monkeypatch.setenv("ODPD_COMM", "bogus")
    with pytest.raises(ValueError):
        odist.comm_candidates("nccl", True)
