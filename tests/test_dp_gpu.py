"""Data-parallel step on the GPU (§8e).  (1) The library-owned RCCL communicator (csrc/comm.hip) on this one-GPU box: a world of one
rank — librccl is loaded, a communicator initialised, the all-reduce enqueued on the step's stream, and the sharded native epoch loop
(odpd_train_epoch_dp) reproduces the single-process loop bit for bit.  (2) Two ranks sharing the GPU over gloo: the all-reduced sum of
the two HIP shard gradients (each normalised by the GLOBAL element count) equals the single-process HIP gradient of the whole batch,
and both replicas hold bit-identical parameters after the step."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stream(n, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = (torch.rand(n, 2, device="cuda", generator=g) - 0.5) * 1.4
    x = x + 0.05 * torch.sign(x)
    y = x * (1.0 - 0.2 * (x * x).sum(-1, keepdim=True)) + 0.05 * torch.roll(x, 1, 0)
    return x.contiguous(), y.contiguous()


class _Loader:
    """the attributes FusedAdamW.train_epoch reads from project.DeviceFrameLoader"""

    def __init__(self, x, y, T, batch, seed):
        self.x, self.y, self.frame_length, self.stride, self.batch_size = x, y, T, 1, batch
        self.n = x.shape[0] - T + 1
        self._order = torch.randperm(self.n, generator=torch.Generator().manual_seed(seed)).cuda()

    def epoch_order(self):
        return self._order


@pytest.mark.parametrize("bb,H,batch", [("dgru", 13, 64), ("gru", 11, 37), ("vdlstm", 13, 48)])
def test_rccl_communicator_of_one_rank_and_the_sharded_epoch_loop(bb, H, batch, monkeypatch):
    from opendpd_amd import CoreModel, _lib, dist as odist
    from opendpd_amd.train_funcs import FusedAdamW
    lib = _lib.load()
    x, y = _stream(1000, 3)
    T = 50
    res = []
    for native in (False, True):
        monkeypatch.setenv("ODPD_NATIVE_COMM", "1" if native else "0")
        odist._native = None
        torch.manual_seed(0)
        net = CoreModel(2, H, 1, bb).cuda()
        opt = FusedAdamW(net, lr=1e-3)
        loader = _Loader(x, y, T, batch, seed=5)
        assert opt.can_run_epoch(loader)
        assert (opt.native_comm() is not None) == native
        losses = opt.train_epoch(loader, "l2", 200.0)
        if native:      # the collective itself, on a buffer: a sum over one rank
            buf = torch.arange(10, dtype=torch.float32, device="cuda")
            opt.native_comm().allreduce_sum_(buf)
            assert torch.equal(buf.cpu(), torch.arange(10, dtype=torch.float32))
        res.append((losses.cpu().numpy(), net.backbone.flat_params().cpu().numpy().copy()))
    odist._native = None
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    lo, hi = C.c_int64(), C.c_int64()
    got = []
    for r in range(3):
        lib.odpd_shard_range(157, r, 3, C.byref(lo), C.byref(hi))
        got.append((lo.value, hi.value))
    assert got == [odist.shard_range(157, r, 3) for r in range(3)] == [(0, 53), (53, 105), (105, 157)]


_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import numpy as np, torch
from opendpd_amd import CoreModel, dist as odist
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
rank, _, world = odist.env_world()
torch.cuda.set_device(0)
odist.init("gloo")
bb, H, B, T = {bb!r}, {H}, {B}, {T}
g = torch.Generator().manual_seed(7)
x = ((torch.rand(B, T, 2, generator=g) - 0.5) * 1.4)
x = (x + 0.05 * torch.sign(x)).cuda()
t = (torch.rand(B, T, 2, generator=g) - 0.5).cuda()
torch.manual_seed(0)
net = CoreModel(2, H, 1, bb).cuda()
opt = FusedAdamW(net, lr=1e-3)
lo, hi = odist.shard_range(B, rank, world)
loss = fused_train_step(opt, x[lo:hi].contiguous(), t[lo:hi].contiguous(), "l2", 200.0, global_count=B * T * 2)
torch.cuda.synchronize()
np.savez({out!r} + f"_{{rank}}.npz", grad=opt.grad.cpu().numpy(), params=net.backbone.flat_params().cpu().numpy(), loss=float(loss.item()))
torch.distributed.destroy_process_group()
"""


@pytest.mark.parametrize("bb,H,B", [("dgru", 13, 64), ("dgru", 13, 7), ("deltagru_tcnskip", 15, 33), ("vdlstm", 13, 50)])
def test_two_rank_sum_of_hip_shard_gradients_equals_the_full_batch_gradient(bb, H, B, tmp_path):
    from opendpd_amd import CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    T = 40
    out = str(tmp_path / "r")
    script = tmp_path / "w.py"
    script.write_text(_WORKER.format(root=ROOT, bb=bb, H=H, B=B, T=T, out=out))
    env = dict(os.environ, ODPD_NATIVE_COMM="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29543", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ranks = [dict(np.load(out + f"_{k}.npz")) for k in range(2)]
    # the single-process step on the whole batch
    g = torch.Generator().manual_seed(7)
    x = ((torch.rand(B, T, 2, generator=g) - 0.5) * 1.4)
    x = (x + 0.05 * torch.sign(x)).cuda()
    t = (torch.rand(B, T, 2, generator=g) - 0.5).cuda()
    torch.manual_seed(0)
    net = CoreModel(2, H, 1, bb).cuda()
    opt = FusedAdamW(net, lr=1e-3)
    loss = fused_train_step(opt, x, t, "l2", 200.0)
    torch.cuda.synchronize()
    P = net.backbone.n_flat
    ref_g, ref_p = opt.grad.cpu().numpy(), net.backbone.flat_params().cpu().numpy()
    # after the all-reduce both ranks hold the same buffer: the global-batch gradient (+ the loss sum in column P)
    assert np.array_equal(ranks[0]["grad"], ranks[1]["grad"])
    scale = np.abs(ref_g[:P]).max()
    assert np.abs(ranks[0]["grad"][:P] - ref_g[:P]).max() <= 1e-6 * scale, np.abs(ranks[0]["grad"][:P] - ref_g[:P]).max() / scale
    assert abs(ranks[0]["loss"] - float(loss.item())) <= 1e-6 * abs(float(loss.item()))
    # identical replicas after the step, equal to the single-process result up to the summation order of the two shards
    assert np.array_equal(ranks[0]["params"], ranks[1]["params"])
    assert np.abs(ranks[0]["params"] - ref_p).max() <= 2e-6 * np.abs(ref_p).max()


@pytest.mark.parametrize("dpd_bb,dpd_h,pa_bb,pa_h", [("gru", 15, "gru", 23), ("deltagru_tcnskip", 15, "dgru", 23)])
def test_sharded_cascade_epoch_loop_with_a_communicator_of_one_rank(dpd_bb, dpd_h, pa_bb, pa_h, monkeypatch):
    """train_dpd under the library-owned communicator: odpd_train_epoch_cascade with comm != NULL (shard ranges, global loss count, the
    all-reduce enqueued between row reduction and optimiser) reproduces the single-process cascade epoch bit for bit on a world of one."""
    from opendpd_amd import CascadedModel, CoreModel, dist as odist
    from opendpd_amd.train_funcs import FusedAdamW
    x, y = _stream(1000, 3)
    T = 50
    res = []
    for native in (False, True):
        monkeypatch.setenv("ODPD_NATIVE_COMM", "1" if native else "0")
        odist._native = None
        torch.manual_seed(0)
        net = CascadedModel(dpd_model=CoreModel(2, dpd_h, 1, dpd_bb, thx=0.01, thh=0.05), pa_model=CoreModel(2, pa_h, 1, pa_bb))
        net.freeze_pa_model()
        net = net.cuda()
        opt = FusedAdamW(net, lr=1e-3)
        loader = _Loader(x, y, T, 64, seed=5)
        assert opt.can_run_cascade_epoch(loader)
        assert (opt.native_comm() is not None) == native
        losses = opt.train_epoch_cascade(loader, "l2", 200.0)
        res.append((losses.cpu().numpy(), net.dpd_model.backbone.flat_params().cpu().numpy().copy()))
    odist._native = None
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
