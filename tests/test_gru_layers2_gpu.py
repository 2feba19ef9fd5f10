"""gru / qgru / qgru_amp1 / lstm with TWO recurrent layers (nn.GRU / nn.LSTM num_layers = 2: backbones/gru.py:17-21, lstm.py:17-21;
`--PA_num_layers 2`), hidden <= 32, on csrc/gru_layers2.hip, lstm_layers2.hip (both layers in one wave, time-skewed by one step): against the
vectors the REFERENCE produced for a two-layer gru and a two-layer lstm (tests/golden/wide_gru_h12_l2.npz, wide_lstm_h10_l2.npz) and, on ragged shapes, against the ATen restatement of the same module (backbones/wide.py: torch's own
nn.GRU, pinned to the reference's vectors by tests/test_wide_cpu.py; the C oracle has no second layer)."""
import contextlib
import warnings

import numpy as np
import pytest
import torch

from tests.golden_util import Fixture, rel_err

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def _aten_only():
    from opendpd_amd.backbones import wide as W
    old = W.TWO_LAYER_KERNELS
    W.TWO_LAYER_KERNELS = ()
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            yield
    finally:
        W.TWO_LAYER_KERNELS = old


@pytest.mark.parametrize("name", ["wide_gru_h12_l2", "wide_lstm_h10_l2", "wide_dgru_h13_l2"])
def test_reference_fixture_of_a_two_layer_model(name):
    from opendpd_amd import CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    fx = Fixture(name)
    m = fx.meta
    assert m["num_layers"] == 2
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        net = CoreModel(2, m["hidden"], 2, m["backbone"])
    assert net.backbone.native
    net.load_state_dict({k: torch.from_numpy(fx["sd/" + k]) for k in fx.keys("sd")})
    net = net.cuda()
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_(True)
    t = torch.from_numpy(fx["tgt"]).cuda()
    y = net(x)
    assert rel_err(y.detach().cpu().numpy(), fx["y"]) < 2e-5
    loss = torch.nn.functional.mse_loss(y, t)
    assert abs(loss.item() - fx["losses"][0]) < 1e-5 * max(1.0, fx["losses"][0])
    loss.backward()
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), fx["g/" + k]) < 3e-4, k
    assert rel_err(x.grad.cpu().numpy(), fx["gx"]) < 3e-4
    opt = FusedAdamW(net, lr=m["lr"])
    fused_train_step(opt, x.detach(), t, "l2", m["clip"])
    for k, p in net.named_parameters():
        assert rel_err(p.detach().cpu().numpy(), fx["p1/" + k]) < 3e-5, k


@pytest.mark.parametrize("bb,H", [("gru", 8), ("gru", 17), ("gru", 32), ("qgru", 10), ("qgru_amp1", 23), ("lstm", 9), ("lstm", 20), ("lstm", 32),
                                  ("dgru", 7), ("dgru", 13), ("dgru", 24), ("dgru", 32)])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (4, 63), (4, 64), (5, 70), (2, 200), (1100, 7)])
def test_against_the_aten_restatement_on_ragged_shapes(bb, H, B, T):
    from opendpd_amd import CoreModel
    torch.manual_seed(H * 100 + B + T)
    net = CoreModel(2, H, 2, bb).cuda()
    assert net.backbone.native
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    with _aten_only():
        ref = CoreModel(2, H, 2, bb)
    assert not ref.backbone.native
    ref.load_state_dict({k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
    ref = ref.double()
    rng = np.random.RandomState(B * 7 + T)
    amp, ph = 0.05 + 0.85 * rng.rand(B, T, 1), 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    dy = rng.randn(B, T, 2).astype(np.float32)
    xr = torch.from_numpy(x).double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(torch.from_numpy(dy).double())
    with torch.no_grad():
        assert rel_err(net(torch.from_numpy(x).cuda()).cpu().numpy(), yr.detach().numpy()) < 2e-5      # inference: no records
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    y = net(xt)
    y.backward(torch.from_numpy(dy).cuda())
    assert rel_err(y.detach().cpu().numpy(), yr.detach().numpy()) < 2e-5
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert rel_err(p.grad.cpu().numpy(), q.grad.numpy()) < 2e-4, k
    assert rel_err(xt.grad.cpu().numpy(), xr.grad.numpy()) < 2e-4
    # dL/dx alone (a frozen two-layer PA in front of the loss)
    for p in net.parameters():
        p.requires_grad_(False)
    xt2 = torch.from_numpy(x).cuda().requires_grad_(True)
    net(xt2).backward(torch.from_numpy(dy).cuda())
    assert rel_err(xt2.grad.cpu().numpy(), xr.grad.numpy()) < 2e-4


def test_entry_points_written_for_one_layer_refuse_the_descriptor():
    import ctypes as C
    from opendpd_amd import CoreModel, _lib
    lib = _lib.load()
    net = CoreModel(2, 8, 2, "gru").cuda()
    d = net.backbone.desc
    assert d.flags & _lib.FLAG_TWO_LAYERS
    assert int(lib.odpd_param_count(C.byref(d))) == net.backbone.n_flat == 3 * 8 * 2 + 3 * 64 + 48 + 2 * 3 * 64 + 48 + 16 + 2
    assert int(lib.odpd_partial_rows(C.byref(d), 64, 50, 1)) < 0            # no fused step
    assert int(lib.odpd_frozen_loss_rows(C.byref(d), 64, 50)) < 0
    assert int(lib.odpd_train_workspace_floats(C.byref(d), 64, 50)) < 0
    d3 = _lib.ModelDesc(_lib.BACKBONE_IDS["vdlstm"], 8, 0.0, 0.0, 0, 0, _lib.FLAG_TWO_LAYERS)
    assert int(lib.odpd_param_count(C.byref(d3))) < 0                       # vdlstm: one layer only


@pytest.mark.parametrize("dpd_bb,H", [("lstm", 12), ("gru", 11), ("dgru", 9)])
def test_two_layer_dpd_in_train_dpd_takes_the_chained_launches(dpd_bb, H):
    """`--DPD_backbone lstm --DPD_num_layers 2` in front of a gru / dgru PA.  The one-launch cascade step only knows
    one-layer parameter layouts: odpd_cascade_rows must refuse the two-layer descriptor (it used to admit the lstm through its backbone id
    and ran the ONE-layer layout over the two-layer buffer), and the chained step's loss / DPD gradient must equal the ATen restatement's."""
    import ctypes as C
    from opendpd_amd import CascadedModel, CoreModel, _lib
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    lib = _lib.load()
    torch.manual_seed(H)
    B, T = 24, 40
    dpd, pa = CoreModel(2, H, 2, dpd_bb), CoreModel(2, 11, 1, "dgru")
    assert dpd.backbone.native
    net = CascadedModel(dpd_model=dpd, pa_model=pa)
    net.freeze_pa_model()
    net = net.cuda()
    assert int(lib.odpd_cascade_rows(C.byref(dpd.backbone.desc), C.byref(pa.backbone.desc), B, T)) < 0
    rng = np.random.RandomState(H)
    x = (rng.uniform(0.1, 0.8, (B, T, 2)) * rng.choice([-1.0, 1.0], (B, T, 2))).astype(np.float32)
    t = (0.5 * rng.randn(B, T, 2)).astype(np.float32)
    with _aten_only():
        ref_dpd = CoreModel(2, H, 2, dpd_bb)
    ref_dpd.load_state_dict(dpd.state_dict())
    ref_pa = CoreModel(2, 11, 1, "dgru")
    ref_pa.load_state_dict(pa.state_dict())
    xr = torch.from_numpy(x)
    with _aten_only():
        loss_ref = torch.nn.functional.mse_loss(ref_pa.cuda()(ref_dpd.cuda()(xr.cuda())), torch.from_numpy(t).cuda())
        loss_ref.backward()
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    lg = fused_train_step(opt, torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), "l2", 0.0)
    assert abs(lg.item() - loss_ref.item()) < 2e-5 * max(1.0, loss_ref.item())
    got = opt.grad[:-4].cpu().numpy()
    want = np.concatenate([p.grad.detach().cpu().numpy().reshape(-1) for p in ref_dpd.parameters()])
    assert rel_err(got, want) < 3e-4
