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
