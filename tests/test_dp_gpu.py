"""Data-parallel step on the GPU (§8e).
(1) One rank, every library-owned communicator (one-shot exchange over hipIpc slots, over host shared memory, RCCL): the collective is
    really enqueued — ncclAllReduce for a world of one too — and the sharded native epoch loops (odpd_train_epoch_dp,
    odpd_train_epoch_cascade) reproduce the single-process loops bit for bit.
(2) Two ranks SHARING the GPU, data plane = the one-shot exchange (control plane gloo): the per-step path and the sharded native epoch
    loops with world = 2 — frame_idx + shard offsets, uneven and empty shards, the global loss count, the exchange folded into the
    optimiser kernel — equal the single-process run up to the summation order of the two shards; replicas bit-identical.
(3) torch.distributed.run with the nccl backend on this one GPU: bench.py's N-rank code path (rendezvous, RCCL process group, id
    broadcast, communicator, barriers, rank-0 JSON) as the driver launches it.
(4) Two GPUs (skipped on a one-GPU box): the same comparisons over real RCCL and over the hipIpc exchange between two devices."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import dp_worker as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PORT = [29540]


def _torchrun(nproc, script_args, env, timeout=600):
    _PORT[0] += 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_PORT[0]), *script_args]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return r


def _ranks(specs, tmp_path, world, comm, backend="gloo", share_gpu=True):
    """the specs run one after the other by `world` ranks of ONE launch: [spec][rank] -> result dict"""
    job = dict(specs=specs, backend=backend, share_gpu=share_gpu, out=str(tmp_path / "r"))
    path = tmp_path / "job.json"
    path.write_text(json.dumps(job))
    env = dict(os.environ, ODPD_COMM=comm, ODPD_XCHG_TIMEOUT_MS="20000")
    env.pop("ODPD_NATIVE_COMM", None)
    _torchrun(world, [os.path.join(ROOT, "tests", "dp_worker.py"), str(path)], env)
    return [[dict(np.load(job["out"] + f"_{i}_{k}.npz")) for k in range(world)] for i in range(len(specs))]


def _single(spec, monkeypatch):
    from opendpd_amd import dist as odist
    monkeypatch.setenv("ODPD_COMM", "torch")
    odist.reset_native_comm()
    return W.run(spec, torch.device("cuda", 0))


def _check(ranks, ref, comm, what=""):
    P = len(ref["params"])
    for r in ranks:
        assert str(r["comm"]) in (comm if isinstance(comm, tuple) else (comm,)) and int(r["errors"]) == 0, (what, str(r["comm"]), int(r["errors"]))
    # every rank holds the same bits: reduced buffer, per-step global losses, parameters
    for k in ("grad", "losses", "params"):
        for r in ranks[1:]:
            assert np.array_equal(ranks[0][k], r[k]), (what, k)
    # ... equal to the single-process run up to the summation order of the shards
    k = 1 + len(ranks) // 4       # (more shards: more re-associated partial sums in the gradient)
    assert np.abs(ranks[0]["losses"] - ref["losses"]).max() <= k * 2e-6 * np.abs(ref["losses"]).max(), what
    assert np.abs(ranks[0]["params"] - ref["params"]).max() <= k * 3e-6 * np.abs(ref["params"]).max(), what
    if len(ref["losses"]) == 1:      # one step: the all-reduced gradient itself
        scale = np.abs(ref["grad"][:P]).max()
        assert np.abs(ranks[0]["grad"][:P] - ref["grad"][:P]).max() <= k * 1e-6 * scale, what


def _name(s):
    return f"{s['mode']}-{s['bb']}{s['H']}-B{s['B']}"


# ---- (1) one rank -------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("comm", ["xchg", "xchg_shm", "rccl"])
@pytest.mark.parametrize("spec", [dict(mode="epoch", bb="dgru", H=13, B=64), dict(mode="epoch", bb="gru", H=11, B=37), dict(mode="epoch", bb="vdlstm", H=13, B=48),
                                  dict(mode="cascade_epoch", bb="gru", H=15, pa_bb="gru", pa_H=23, B=64),
                                  dict(mode="cascade_epoch", bb="deltagru_tcnskip", H=15, pa_bb="dgru", pa_H=23, B=64)],
                         ids=lambda s: f"{s['mode']}-{s['bb']}{s['H']}")
def test_communicator_of_one_rank_and_the_sharded_epoch_loops(spec, comm, monkeypatch):
    from opendpd_amd import _lib, dist as odist
    spec = dict(spec, T=50, n=1000)
    ref = _single(spec, monkeypatch)
    monkeypatch.setenv("ODPD_COMM", comm)
    odist.reset_native_comm()
    got = W.run(spec, torch.device("cuda", 0))
    assert str(got["comm"]) == comm and int(got["errors"]) == 0
    assert np.array_equal(got["losses"], ref["losses"]) and np.array_equal(got["params"], ref["params"])
    # the collective itself, on a buffer: a sum over one rank — through ncclAllReduce / the exchange kernel, not around them
    native = odist.native_comm()
    assert native is not None and _lib.load().odpd_comm_kind(native.handle) == {"rccl": 0, "xchg": 1, "xchg_shm": 2}[comm]
    buf = torch.arange(3000, dtype=torch.float32, device="cuda")
    native.allreduce_sum_(buf)
    assert torch.equal(buf.cpu(), torch.arange(3000, dtype=torch.float32))
    odist.reset_native_comm()


def test_shard_ranges_of_the_library_match_the_host_side():
    from opendpd_amd import _lib, dist as odist
    lib = _lib.load()
    lo, hi = C.c_int64(), C.c_int64()
    got = []
    for r in range(3):
        lib.odpd_shard_range(157, r, 3, C.byref(lo), C.byref(hi))
        got.append((lo.value, hi.value))
    assert got == [odist.shard_range(157, r, 3) for r in range(3)] == [(0, 53), (53, 105), (105, 157)]


def test_fused_and_standalone_exchange_agree(monkeypatch):
    """odpd_set_tuning("xchg_fused", 0): the exchange as its own launch in front of the optimiser kernel — same bits as the prologue form"""
    from opendpd_amd import _lib, dist as odist
    spec = dict(mode="epoch", bb="dgru", H=13, B=64, T=50, n=600)
    monkeypatch.setenv("ODPD_COMM", "xchg")
    out = []
    for fused in (1, 0):
        odist.reset_native_comm()
        _lib.load().odpd_set_tuning(b"xchg_fused", fused)
        out.append(W.run(spec, torch.device("cuda", 0)))
    _lib.load().odpd_set_tuning(b"xchg_fused", 1)
    odist.reset_native_comm()
    assert np.array_equal(out[0]["losses"], out[1]["losses"]) and np.array_equal(out[0]["params"], out[1]["params"])


# ---- (2) two ranks sharing the GPU over the one-shot exchange -------------------------------------------------------------------------
_STEP_SPECS = [dict(mode="step", bb="dgru", H=13, B=64), dict(mode="step", bb="dgru", H=13, B=7), dict(mode="step", bb="deltagru_tcnskip", H=15, B=33),
               dict(mode="step", bb="vdlstm", H=13, B=50), dict(mode="step", bb="dgru", H=13, B=1, steps=3),
               dict(mode="cascade_step", bb="deltagru_tcnskip", H=15, pa_bb="dgru", pa_H=23, B=33)]
_EPOCH_SPECS = [dict(mode="epoch", bb="dgru", H=13, B=64, n=600), dict(mode="epoch", bb="gru", H=11, B=37, n=500), dict(mode="epoch", bb="vdlstm", H=13, B=49, n=500),
                dict(mode="epoch", bb="dgru", H=13, B=50, n=140),        # 101 frames: the last global batch is ONE frame -> rank 1's shard is empty
                dict(mode="cascade_epoch", bb="gru", H=15, pa_bb="gru", pa_H=23, B=64, n=500),
                dict(mode="cascade_epoch", bb="deltagru_tcnskip", H=15, pa_bb="dgru", pa_H=23, B=33, n=400)]


@pytest.mark.parametrize("comm", ["xchg", "xchg_shm"])
def test_two_ranks_on_one_gpu_equal_the_single_process_run(comm, tmp_path, monkeypatch):
    specs = _STEP_SPECS + _EPOCH_SPECS
    if comm == "xchg_shm":      # the shared-memory transport runs the same kernels: a subset of the cases covers it
        specs = [_STEP_SPECS[0], _EPOCH_SPECS[0], _EPOCH_SPECS[3], _EPOCH_SPECS[5]]
    specs = [dict(s, T=40) for s in specs]
    got = _ranks(specs, tmp_path, 2, comm)
    for spec, ranks in zip(specs, got):
        _check(ranks, _single(spec, monkeypatch), comm, _name(spec))


def test_two_ranks_over_gloo_fall_back_to_torch_distributed(tmp_path, monkeypatch):
    """ODPD_COMM=auto on a gloo group: no library-owned communicator, the per-step torch.distributed all-reduce — same contract"""
    spec = dict(mode="step", bb="dgru", H=13, B=64, T=40)
    (ranks,) = _ranks([spec], tmp_path, 2, "auto")
    _check(ranks, _single(spec, monkeypatch), "torch")


# ---- (3) the driver's launch line on one GPU, RCCL process group -------------------------------------------------------------------
@pytest.mark.parametrize("comm", ["rccl", "xchg"])
def test_bench_under_torchrun_with_the_nccl_backend_on_one_rank(comm):
    env = dict(os.environ, ODPD_COMM=comm, ODPD_BENCH_NO_SUSTAINED="1")
    env.pop("ODPD_NATIVE_COMM", None)
    r = _torchrun(1, [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "4096", "--no-cascade",
                      "--no-cpu-baseline", "--ref-batch", "0"], env)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and np.isfinite(line["config"]["loss"])
    assert line["collective"]["kind"] == comm and line["collective"]["process_group"] == "nccl"


# ---- (4) two GPUs: RCCL and the hipIpc exchange between devices ------------------------------------------------------------------------
_TWO_GPU_SPECS = [dict(mode="step", bb="dgru", H=13, B=64), dict(mode="epoch", bb="dgru", H=13, B=64, n=600), dict(mode="epoch", bb="vdlstm", H=13, B=49, n=500),
                  dict(mode="epoch", bb="dgru", H=13, B=50, n=140),
                  dict(mode="cascade_epoch", bb="deltagru_tcnskip", H=15, pa_bb="dgru", pa_H=23, B=33, n=400)]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("comm", ["auto", "rccl", "xchg"])
def test_two_gpus_equal_the_single_process_run(comm, tmp_path, monkeypatch):
    """dgru, cfg 4's vdlstm and cfg 3's cascade on two devices: all-reduced result == single process, replicas bit-identical"""
    specs = [dict(s, T=40) for s in _TWO_GPU_SPECS]
    got = _ranks(specs, tmp_path, 2, comm, backend="nccl", share_gpu=False)
    for spec, ranks in zip(specs, got):
        _check(ranks, _single(spec, monkeypatch), ("xchg", "rccl") if comm == "auto" else comm, _name(spec))


def test_a_peer_that_never_arrives_ends_in_nan_and_an_error_count_not_in_a_hung_gpu():
    """rank 0 of a two-rank shared-memory communicator whose rank 1 does not exist: the exchange gives up after the communicator's time-out,
    poisons the sum with NaN (so the step's loss and parameters say what happened) and counts the event"""
    from opendpd_amd import _lib
    lib = _lib.load()
    comm, h = C.c_void_p(), (C.c_ubyte * 64)()
    name = f"/odpd_xchg_test_{os.getpid()}".encode()
    assert lib.odpd_xchg_create(2, 0, name, C.byref(comm), C.cast(h, C.c_void_p)) == 0
    try:
        assert lib.odpd_xchg_connect(comm, None) == 0 and lib.odpd_xchg_unlink(comm) == 0
        assert lib.odpd_comm_set_timeout_ms(comm, 200) == 0
        buf = torch.arange(1045, dtype=torch.float32, device="cuda")
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        assert lib.odpd_comm_allreduce_sum(_lib.stream_ptr(), comm, _lib.ptr(buf), buf.numel()) == 0
        t1.record()
        torch.cuda.synchronize()
        assert 150.0 <= t0.elapsed_time(t1) <= 2000.0
        assert bool(torch.isnan(buf).all()) and lib.odpd_comm_errors(comm) == 1      # one EXCHANGE timed out (not: 1045 elements)
    finally:
        lib.odpd_comm_destroy(comm)


@pytest.mark.parametrize("comm", ["xchg", "xchg_shm"])
def test_eight_ranks_on_one_gpu_equal_the_single_process_run(comm, tmp_path, monkeypatch):
    """The driver's N = 8 shape on the one-GPU box: eight processes share the device, data plane = the one-shot exchange with a world of
    eight (slot rows of eight sources, both parities, rank-order sums), shards of 8 / 4 / 1 / 0 frames per rank, whole sharded epochs from
    the native loops — against the single-process run."""
    specs = [dict(mode="step", bb="dgru", H=13, B=64, steps=3), dict(mode="step", bb="dgru", H=13, B=5, steps=2),       # B = 5: ranks 5..7 hold empty shards
             dict(mode="epoch", bb="dgru", H=13, B=64, n=360), dict(mode="epoch", bb="vdlstm", H=13, B=37, n=200)]
    if comm == "xchg":
        specs.append(dict(mode="cascade_epoch", bb="deltagru_tcnskip", H=15, pa_bb="dgru", pa_H=23, B=32, n=170))
    specs = [dict(s, T=40) for s in specs]
    got = _ranks(specs, tmp_path, 8, comm)
    for spec, ranks in zip(specs, got):
        ref = _single(spec, monkeypatch)
        assert len(ranks) == 8
        _check(ranks, ref, comm, _name(spec))
