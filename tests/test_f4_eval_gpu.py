"""GPU parity of the one-sequence-per-workgroup EVALUATION kernels of bojanet, apnrru, dvrjanet and mcldnn (boj_gp_eval_kernel,
apn_gp_eval_kernel, dvr_gp_eval_kernel, mcl_gp_eval_kernel: the reference's net_eval / run_dpd shapes, train_funcs.py:57-90 — a few
sequences of thousands of steps, forward only): against the CPU oracle and against the 16-sequences-per-wave forward kernels, across
the 256-step chunk boundaries of the kernels, at the shortest legal frames, and for more sequences than one round of workgroups."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.golden_util import rel_err

pytestmark = pytest.mark.gpu

CASES = [("bojanet", 12, {}), ("bojanet", 5, {}), ("bojanet", 16, {}), ("apnrru", 8, {}), ("apnrru", 14, {}), ("apnrru", 3, {}),
         ("dvrjanet", 12, {"num_dvr_units": 3}), ("dvrjanet", 16, {"num_dvr_units": 8}), ("dvrjanet", 5, {"num_dvr_units": 4}),
         ("mcldnn", 8, {}), ("mcldnn", 3, {}), ("mcldnn", 16, {})]
MIN_T = {"bojanet": 15, "apnrru": 15, "dvrjanet": 1, "mcldnn": 4}


def _net(bb, H, kw):
    from opendpd_amd import CoreModel
    torch.manual_seed(H * 7 + len(bb))
    net = CoreModel(2, H, 1, bb, **kw)
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
        if bb == "bojanet":
            net.backbone.fir_I.weight.mul_(4.0)
            net.backbone.fir_Q.weight.mul_(4.0)
        if bb == "apnrru":
            net.backbone.rru.Z.uniform_(-0.6, 0.6)
        if bb == "dvrjanet":        # (tests/test_dvrjanet_gpu.py: keeps the recurrence out of its chaotic regime)
            net.backbone.W_ax.weight.mul_(1.5)
            net.backbone.cs.mul_(min(1.0, 1.5 / float(net.backbone.cs.abs().sum())))
    return net.cuda().eval()


def _iq(rng, B, T):
    amp, ph = 0.05 + 0.85 * rng.rand(B, T, 1), 2 * np.pi * rng.rand(B, T, 1)
    return np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)


@pytest.mark.parametrize("bb,H,kw", CASES)
@pytest.mark.parametrize("B,T", [(1, 0), (2, 255), (1, 256), (3, 257), (1, 513), (2, 1300), (600, 40)])
def test_evaluation_kernels(bb, H, kw, B, T):
    from opendpd_amd import _lib
    from oracle.oracle import Oracle, make_model
    lib = _lib.load()
    T = T or MIN_T[bb]
    net = _net(bb, H, kw)
    rng = np.random.RandomState(B * 31 + T)
    x = _iq(rng, B, T)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    m = make_model(bb, H, bits_w=kw.get("num_dvr_units", 0))
    yo, _ = Oracle("f32").forward(m, p, x)
    y64, _ = Oracle("f64").forward(m, p.astype(np.float64), x.astype(np.float64))
    tol = max(2e-5, 6 * rel_err(yo, y64))            # the oracle's own fp32-vs-fp64 distance on this draw (long recurrences amplify rounding)
    xt = torch.from_numpy(x).cuda()
    try:
        with torch.no_grad():
            y_eval = net(xt).cpu().numpy()
            lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(0))
            y_s16 = net(xt).cpu().numpy()
    finally:
        lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(-1))
    assert rel_err(y_eval, yo) < tol and rel_err(y_s16, yo) < tol
    assert rel_err(y_eval, y_s16) < tol
    assert B > 512 or T < 2 or not np.array_equal(y_eval, y_s16)          # two kernels (batches past two rounds of workgroups stay on the S16 one)


def test_full_evaluation_length():
    """one sequence of the reference's evaluation length (19 662 samples) against the fp64 oracle"""
    from oracle.oracle import Oracle, make_model
    for bb, H, kw in (("bojanet", 12, {}), ("apnrru", 8, {}), ("mcldnn", 8, {})):
        net = _net(bb, H, kw)
        x = _iq(np.random.RandomState(3), 1, 19662)
        p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
        m = make_model(bb, H)
        yo, _ = Oracle("f32").forward(m, p, x)
        y64, _ = Oracle("f64").forward(m, p.astype(np.float64), x.astype(np.float64))
        with torch.no_grad():
            y = net(torch.from_numpy(x).cuda()).cpu().numpy()
        assert rel_err(y, y64) < max(2e-5, 6 * rel_err(yo, y64)), bb
