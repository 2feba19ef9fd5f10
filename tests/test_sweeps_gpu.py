"""Randomised launch-flavour sweeps as tests (they were tools until r02): every instantiation that runs at a raised occupancy or spills
registers is compared with the oracle on every launch path that can select it, so that a compiler bump cannot silently change one.
  * csrc/gru_s16n.hip, hidden 17..32, all four feature sets: frozen-PA cascade step, frozen backward (dL/dx only), weight gradients,
    both together, fused train step (tools/s16n_crosscheck.py)
  * csrc/qat_s16.hip forced at small batches for the kinds the row-rotated QAT kernels also serve: forward, weight gradients only,
    dL/dx only, both."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest
import torch

from tests.golden_util import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def force_s16():
    from opendpd_amd import _lib
    lib = _lib.load()
    assert lib.odpd_set_tuning(b"s16_min_batch", C.c_int64(0)) == 0
    yield
    lib.odpd_set_tuning(b"s16_min_batch", C.c_int64(-1))


def _crosscheck():
    spec = importlib.util.spec_from_file_location("s16n_crosscheck", os.path.join(ROOT, "tools", "s16n_crosscheck.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("pbb", ["dgru", "gru", "qgru", "qgru_amp1"])
@pytest.mark.parametrize("ph", list(range(17, 33)))
def test_hidden_17_to_32_launch_flavours_against_the_oracle(force_s16, pbb, ph):
    worst, bad, kinks, n = _crosscheck().check(pbb, ph)
    assert n == 7 and not bad, bad
    assert len(kinks) <= 2, kinks           # draws on a relu kink / next to the origin are classified against the fp64 oracle, not counted
    assert max(worst) < 3e-4, worst


class _Proj:
    quant = True
    pretrained_model = ""
    n_bits_w = n_bits_a = 8


@pytest.mark.parametrize("bb", ["qgru", "qgru_amp1"])
@pytest.mark.parametrize("H,B,T", [(10, 3, 5), (16, 7, 33), (6, 66, 63), (13, 5, 200), (1, 17, 9)])
def test_qat_s16_flavours_where_the_row_rotated_kernels_would_run(force_s16, bb, H, B, T):
    from opendpd_amd import CoreModel
    from opendpd_amd.quant import get_quant_model
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H + B + T)
    q = get_quant_model(_Proj, CoreModel(2, H, 1, bb)).cuda()
    rng = np.random.RandomState(B + T)
    amp = 0.05 + 0.85 * rng.rand(B, T, 1)
    ph = 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    dy = rng.randn(B, T, 2).astype(np.float32)
    o = Oracle("f32")
    m = make_model(bb, H, bits_w=8, bits_a=8)
    p = np.concatenate([v.detach().cpu().numpy().reshape(-1) for v in q.parameters()])
    yo = o.qat_forward(m, p, x)
    go, dxo = o.qat_backward(m, p, x, dy, need_dx=True)
    grads = lambda: np.concatenate([(v.grad if v.grad is not None else torch.zeros_like(v)).cpu().numpy().reshape(-1) for v in q.parameters()])
    q.train()
    # weight gradients only
    y = q(torch.from_numpy(x).cuda())
    assert np.array_equal(y.detach().cpu().numpy(), yo)
    y.backward(torch.from_numpy(dy).cuda())
    assert rel_err(grads(), go) < 3e-5
    # both
    for v in q.parameters():
        v.grad = None
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    q(xt).backward(torch.from_numpy(dy).cuda())
    assert rel_err(grads(), go) < 3e-5 and rel_err(xt.grad.cpu().numpy(), dxo) < 3e-5
    # dL/dx only (frozen model)
    for v in q.parameters():
        v.requires_grad_(False)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    q(xt).backward(torch.from_numpy(dy).cuda())
    assert rel_err(xt.grad.cpu().numpy(), dxo) < 3e-5
