"""backbones/wide.py (ATen restatements used OUTSIDE the HIP kernels' envelope) against reference-generated fixtures
(oracle/gen_golden.py wide: two layers / hidden sizes the kernels do not reach) and against the C oracle.  CPU only."""
import warnings

import numpy as np
import pytest
import torch

from tests.golden_util import Fixture, rel_err

CASES = ["wide_gru_h12_l2", "wide_dgru_h40", "wide_lstm_h10_l2", "wide_vdlstm_h36", "wide_qgru_amp1_h34", "wide_deltagru_h34",
         "wide_tres_h33", "wide_pgjanet_h18", "wide_tcnn_c66", "wide_gru_h48", "wide_dgru_h64", "wide_lstm_h40", "wide_dgru_h13_l2"]


import contextlib


@contextlib.contextmanager
def _aten_only():
    """the GRU family, lstm, vdlstm and the delta-GRU backbones of 33 .. 64 units are kernel-backed since r04 (csrc/*_wide.hip; GPU tests: test_gru_wide_gpu.py, with these
    same fixtures): here the registry is told to build their ATen restatements all the same — they still serve hidden > 64 and two layers, and
    stay pinned to the reference's vectors."""
    from opendpd_amd.backbones import wide as W
    old, old2 = dict(W.KERNEL_HIDDEN_LIMIT), W.TWO_LAYER_KERNELS
    W.KERNEL_HIDDEN_LIMIT.update(gru=32, dgru=32, qgru=32, qgru_amp1=32, lstm=32, vdlstm=32, deltagru=32, deltagru_tcnskip=32, pgjanet=16)
    W.TWO_LAYER_KERNELS = ()
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            yield
    finally:
        W.KERNEL_HIDDEN_LIMIT.clear()
        W.KERNEL_HIDDEN_LIMIT.update(old)
        W.TWO_LAYER_KERNELS = old2


def _model(fx, seed=None):
    from opendpd_amd import CoreModel
    m = fx.meta
    if seed is not None:
        torch.manual_seed(seed)
    with _aten_only():
        net = CoreModel(2, m["hidden"], m["num_layers"], m["backbone"], thx=m["thx"], thh=m["thh"])
    assert net.backbone.native is False
    return net


@pytest.mark.parametrize("name", CASES)
def test_state_dict_and_init_match_reference(name):
    """same keys / shapes, and — built from the same seed — the same initial values as the reference's constructor"""
    fx = Fixture(name)
    net = _model(fx, seed=0)
    sd = net.state_dict()
    assert list(sd.keys()) == fx.keys("sd")
    for k in fx.keys("sd"):
        assert tuple(sd[k].shape) == fx["sd/" + k].shape, k
        # same generator consumption; orthogonal_'s QR may round differently in the last ulp across LAPACK thread counts
        assert np.allclose(sd[k].numpy(), fx["sd/" + k], rtol=0, atol=2e-6), k
    assert sum(p.numel() for p in net.parameters()) == fx.meta["n_param"]


@pytest.mark.parametrize("name", CASES)
def test_forward_loss_grads_and_one_step_follow_reference(name):
    fx = Fixture(name)
    net = _model(fx)
    net.load_state_dict({k: torch.from_numpy(fx["sd/" + k]) for k in fx.keys("sd")})
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    if hasattr(net.backbone, "set_debug"):
        net.backbone.set_debug(1)
    y = net(x)
    assert rel_err(y.detach().numpy(), fx["y"]) < 2e-5
    loss = torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"]))
    assert abs(loss.item() - fx["losses"][0]) < 1e-5 * max(1.0, fx["losses"][0])
    loss.backward()
    for k, p in net.named_parameters():
        assert rel_err(p.grad.numpy(), fx["g/" + k]) < 3e-4, k
    assert rel_err(x.grad.numpy(), fx["gx"]) < 3e-4
    if "stats" in fx.d:
        st = net.backbone.statistics
        got = np.array([st["num_dx_zeros"], st["num_dx_numel"], st["num_dh_zeros"], st["num_dh_numel"]])
        assert np.array_equal(got, fx["stats"]), (got, fx["stats"])
        assert "HW_PARAM" in net.backbone.get_temporal_sparsity()
    # one clip + AdamW step through the project's optimiser selection (torch.optim for a non-native model)
    from opendpd_amd.project import Project
    opt = torch.optim.AdamW(net.parameters(), lr=fx.meta["lr"])
    torch.nn.utils.clip_grad_norm_(net.parameters(), fx.meta["clip"])
    opt.step()
    for k, p in net.named_parameters():
        assert rel_err(p.detach().numpy(), fx["p1/" + k]) < 3e-5, k
    assert Project is not None


@pytest.mark.parametrize("bb,H", [("gru", 40), ("dgru", 48), ("lstm", 40), ("vdlstm", 36), ("deltagru", 40), ("deltagru_tcnskip", 40),
                                  ("pgjanet", 24)])    # tcnn beyond 64 channels: reference fixture wide_tcnn_c66 (the oracle stops at 64)
def test_against_oracle(bb, H):
    """single-layer wide models: outputs, parameter gradients and dL/dx against the C oracle (which stops at hidden 64)"""
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H)
    kw = dict(thx=0.01, thh=0.02) if "delta" in bb else {}
    with _aten_only():
        net = CoreModel(2, H, 1, bb, **kw)
    rng = np.random.RandomState(H)
    amp, ph = 0.05 + 0.85 * rng.rand(3, 19, 1), 2 * np.pi * rng.rand(3, 19, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    dy = rng.randn(3, 19, 2).astype(np.float32)
    xt = torch.from_numpy(x).requires_grad_(True)
    y = net(xt)
    y.backward(torch.from_numpy(dy))
    o = Oracle("f32")
    m = make_model(bb, H, kw.get("thx", 0), kw.get("thh", 0))
    p = np.concatenate([q.detach().numpy().reshape(-1) for q in net.parameters()])
    yo, _ = o.forward(m, p, x)
    go, dxo = o.backward(m, p, x, dy, need_dx="delta" not in bb)
    g = np.concatenate([q.grad.numpy().reshape(-1) for q in net.parameters()])
    assert rel_err(y.detach().numpy(), yo) < 2e-5
    assert rel_err(g, go) < 3e-4
    if dxo is not None:
        assert rel_err(xt.grad.numpy(), dxo) < 3e-4


def test_inside_the_envelope_the_kernels_are_used_and_outside_a_warning_is_raised():
    from opendpd_amd import CoreModel
    assert CoreModel(2, 32, 1, "dgru").backbone.native is True
    assert CoreModel(2, 64, 1, "dgru").backbone.native is True          # 33 .. 64 units: csrc/gru_wide.hip (r04)
    assert CoreModel(2, 16, 1, "pgjanet").backbone.native is True
    assert CoreModel(2, 32, 1, "pgjanet").backbone.native is True            # 17 .. 32 units: csrc/janet_wide.hip (r04)
    with pytest.warns(UserWarning, match="outside the HIP kernels' envelope"):
        net = CoreModel(2, 65, 1, "dgru")
    assert net.backbone.native is False
    with pytest.warns(UserWarning, match="outside the HIP kernels' envelope"):
        assert CoreModel(2, 33, 1, "pgjanet").backbone.native is False
    assert CoreModel(2, 8, 2, "gru").backbone.native is True            # two layers of <= 32 units: csrc/gru_layers2.hip (r04)
    with pytest.warns(UserWarning, match="outside the HIP kernels' envelope"):
        assert CoreModel(2, 8, 2, "vdlstm").backbone.native is False
    with pytest.warns(UserWarning, match="outside the HIP kernels' envelope"):
        assert CoreModel(2, 8, 3, "gru").backbone.native is False


def test_fused_optimiser_declines_wide_models():
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.train_funcs import FusedAdamW
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wide = CoreModel(2, 40, 1, "pgjanet")
        with pytest.raises(TypeError):
            FusedAdamW(wide)
        casc = CascadedModel(dpd_model=CoreModel(2, 8, 1, "dgru"), pa_model=wide)
        casc.freeze_pa_model()
        with pytest.raises(TypeError):
            FusedAdamW(casc)


@pytest.mark.parametrize("name,bb,H", [("wide_gru_h12_l2", "gru", 12), ("wide_lstm_h10_l2", "lstm", 10), ("wide_dgru_h13_l2", "dgru", 13)])
def test_two_layer_kernel_module_has_the_reference_state_dict_and_init(name, bb, H):
    """gru / lstm with num_layers 2 as the kernel-backed module (csrc/gru_layers2.hip, lstm_layers2.hip): same keys, shapes and — from the same seed —
    initial values as the reference's constructor (nn.GRU / nn.LSTM initialise layer after layer; only weight_ih_l0 is re-drawn xavier: gru.py:27-43)"""
    from opendpd_amd import CoreModel
    fx = Fixture(name)
    torch.manual_seed(0)
    net = CoreModel(2, H, 2, bb)
    assert net.backbone.native is True
    sd = net.state_dict()
    assert list(sd.keys()) == fx.keys("sd")
    for k in fx.keys("sd"):
        assert tuple(sd[k].shape) == fx["sd/" + k].shape, k
        assert np.allclose(sd[k].numpy(), fx["sd/" + k], rtol=0, atol=2e-6), k
    assert sum(p.numel() for p in net.parameters()) == fx.meta["n_param"] == net.backbone.n_flat
