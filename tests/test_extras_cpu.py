"""The SURVEY §8 f4 registry names (all HIP-backed inside their envelopes since r02; opendpd_amd/backbones/extras.py keeps their torch
restatements for configurations beyond them), pinned to vectors produced by running the reference (oracle/gen_golden_extras.py): a
seeded construction through the registry reproduces the reference's initial state dict bit for bit (same parameter names, shapes, order
and RNG consumption) with the native class, and the restatement class, loaded with the stored weights, matches the reference's output,
loss, parameter gradients and input gradient to fp32 round-off (1e-5 of the tensor's max magnitude)."""
import numpy as np
import pytest
import torch

from tests.golden_util import Fixture, rel_err

TOL = 1e-5


def _build(bb, H):
    from opendpd_amd import CoreModel
    torch.manual_seed(0)
    return CoreModel(2, H, 1, bb, window_size=4, num_dvr_units=4, thx=0.01, thh=0.05)


def test_fused_optimizer_refuses_non_native_backbones_and_project_falls_back():
    from opendpd_amd.train_funcs import FusedAdamW
    with pytest.warns(UserWarning, match="envelope"):
        net = _build("mcldnn", 20)          # beyond the kernels' envelope: the ATen restatement
    with pytest.raises(TypeError):
        FusedAdamW(net)


def test_gmp_is_native_and_constructs_like_the_reference():
    """gmp left this module for csrc/gmp.hip; its seeded construction still reproduces the reference's state dict."""
    fx = Fixture("extra_gmp_h8")
    net = _build("gmp", 8)
    after = float(torch.rand(1))
    sd = net.state_dict()
    assert list(sd.keys()) == fx.keys("sd")
    assert np.array_equal(sd["backbone.Weight"].numpy(), fx["sd/backbone.Weight"])
    assert after == fx.meta["rng_after_init"]
    assert net.backbone.native is True and sum(p.numel() for p in net.parameters()) == 495


def test_rvtdcnn_is_native_and_constructs_like_the_reference():
    """rvtdcnn left this module for csrc/rvtdcnn.hip (fc_hid_size <= 32); the seeded construction still reproduces the reference's
    state dict and RNG consumption; beyond the envelope the ATen restatement of extras.py serves it, with a warning."""
    fx = Fixture("extra_rvtdcnn_h6")
    net = _build("rvtdcnn", 6)
    after = float(torch.rand(1))
    sd = net.state_dict()
    assert list(sd.keys()) == fx.keys("sd")
    for k in sd:
        assert np.array_equal(sd[k].numpy(), fx["sd/" + k]), k
    assert after == fx.meta["rng_after_init"]
    assert net.backbone.native is True and sum(p.numel() for p in net.parameters()) == fx.meta["n_param"] == 39 * 6 + 32
    with pytest.warns(UserWarning, match="outside"):
        wide = _build("rvtdcnn", 40)
    assert wide.backbone.native is False and sum(p.numel() for p in wide.parameters()) == 39 * 40 + 32
    # the restatement keeps computing what the reference computes (vectors of the H = 6 case)
    from opendpd_amd.backbones.extras import RVTDCNN
    ref = RVTDCNN(fc_hid_size=6)
    ref.load_state_dict({k[len("backbone."):]: torch.from_numpy(fx["sdu/" + k]) for k in fx.keys("sdu")})
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    y = ref(x)
    assert rel_err(y.detach().numpy(), fx["y"]) < TOL
    torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"])).backward()
    assert rel_err(x.grad.numpy(), fx["gx"]) < 10 * TOL


def test_neuraltx_is_native_and_constructs_like_the_reference():
    """neuraltx left this module for the NTX instantiation of csrc/tcnn.hip (<= 64 channels); the seeded construction — including the
    second reset_parameters() the registry issues — still reproduces the reference's state dict and RNG consumption."""
    fx = Fixture("extra_neuraltx_h12")
    net = _build("neuraltx", 12)
    after = float(torch.rand(1))
    sd = net.state_dict()
    assert list(sd.keys()) == fx.keys("sd")
    for k in sd:
        assert np.array_equal(sd[k].numpy(), fx["sd/" + k]), k
    assert after == fx.meta["rng_after_init"]
    assert net.backbone.native is True and sum(p.numel() for p in net.parameters()) == fx.meta["n_param"] == 27 * 12 + 14
    with pytest.warns(UserWarning, match="outside"):
        wide = _build("neuraltx", 70)
    assert wide.backbone.native is False
    from opendpd_amd.backbones.extras import NeuralTX
    ref = NeuralTX(hidden_channels=12)
    ref.load_state_dict({k[len("backbone."):]: torch.from_numpy(fx["sdu/" + k]) for k in fx.keys("sdu")})
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    y = ref(x)
    assert rel_err(y.detach().numpy(), fx["y"]) < TOL
    torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"])).backward()
    assert rel_err(x.grad.numpy(), fx["gx"]) < 10 * TOL


def test_deltajanet_is_native_and_constructs_like_the_reference():
    """deltajanet left this module for the JAN instantiation of csrc/delta_s16.hip (hidden <= 32); the seeded construction still
    reproduces the reference's state dict and RNG consumption; beyond the envelope (or with more layers) the restatement serves it."""
    fx = Fixture("extra_deltajanet_h10")
    net = _build("deltajanet", 10)
    after = float(torch.rand(1))
    sd = net.state_dict()
    assert list(sd.keys()) == fx.keys("sd")
    for k in sd:
        assert np.array_equal(sd[k].numpy(), fx["sd/" + k]), k
    assert after == fx.meta["rng_after_init"]
    assert net.backbone.native is True and sum(p.numel() for p in net.parameters()) == fx.meta["n_param"] == 2 * 100 + 18 * 10 + 2
    assert (net.backbone.thx, net.backbone.thh) == (0.01, 0.05) and (net.backbone.desc.thx, net.backbone.desc.thh) == (0.0, 0.0)
    assert _build("deltajanet", 40).backbone.native is True          # 33 .. 64 units: csrc/deltajanet_wide.hip (r04)
    with pytest.warns(UserWarning, match="outside"):
        wide = _build("deltajanet", 70)
    assert wide.backbone.native is False
    from opendpd_amd.backbones.extras import DeltaJANET
    ref = DeltaJANET(input_size=6, hidden_size=10, output_size=2, num_layers=1, thx=0.01, thh=0.05)
    ref.load_state_dict({k[len("backbone."):]: torch.from_numpy(fx["sdu/" + k]) for k in fx.keys("sdu")})
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    y = ref(x)
    assert rel_err(y.detach().numpy(), fx["y"]) < TOL
    torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"])).backward()
    assert rel_err(x.grad.numpy(), fx["gx"]) < 10 * TOL


def test_dvrjanet_is_native_and_constructs_like_the_reference():
    """dvrjanet left this module for csrc/dvrjanet_s16.hip (hidden <= 16, <= 8 DVR units); the seeded construction still reproduces the
    reference's state dict (the DVR coefficients drawn after the four input / recurrent matrices) and RNG consumption; beyond the
    envelope the restatement serves it, with a warning, and keeps computing what the reference computes."""
    fx = Fixture("extra_dvrjanet_h8")
    net = _build("dvrjanet", 8)
    after = float(torch.rand(1))
    sd = net.state_dict()
    assert list(sd.keys()) == fx.keys("sd")
    for k in sd:
        assert np.array_equal(sd[k].numpy(), fx["sd/" + k]), k
    assert after == fx.meta["rng_after_init"]
    assert net.backbone.native is True and sum(p.numel() for p in net.parameters()) == fx.meta["n_param"] == 4 + 7 * 64 + 7 * 8 + 2
    assert net.backbone.desc.bits_w == 4
    with pytest.warns(UserWarning, match="envelope"):
        wide = _build("dvrjanet", 20)
    assert wide.backbone.native is False
    from opendpd_amd.backbones.extras import DVRJANET
    ref = DVRJANET(hidden_size=8, output_size=2, num_dvr_units=4)
    ref.load_state_dict({k[len("backbone."):]: torch.from_numpy(fx["sdu/" + k]) for k in fx.keys("sdu")})
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    y = ref(x)
    assert rel_err(y.detach().numpy(), fx["y"]) < TOL
    loss = torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"]))
    loss.backward()
    for k, p in ref.named_parameters():
        assert rel_err(p.grad.numpy(), fx["g/backbone." + k]) < 10 * TOL, k
    assert rel_err(x.grad.numpy(), fx["gx"]) < 10 * TOL


@pytest.mark.parametrize("H", [8, 15])
def test_bojanet_is_native_and_constructs_like_the_reference(H):
    """bojanet left this module for csrc/bojanet_s16.hip (hidden <= 16); the seeded construction — the constructor's own draw and the
    registry's second reset_parameters() — still reproduces the reference's state dict and RNG consumption; hidden 17, 18 run the announced ATen restatement."""
    fx = Fixture(f"extra_bojanet_h{H}")
    net = _build("bojanet", H)
    after = float(torch.rand(1))
    sd = net.state_dict()
    assert list(sd.keys()) == fx.keys("sd")
    for k in sd:
        assert np.array_equal(sd[k].numpy(), fx["sd/" + k]), k
    assert after == fx.meta["rng_after_init"]
    assert net.backbone.native is True and sum(p.numel() for p in net.parameters()) == fx.meta["n_param"] == 2 * H * H + 28 * H + 194
    # hidden 17, 18: beyond the kernel's unit tile — the announced ATen restatement (restored in r04), checked against the oracle
    from oracle.oracle import Oracle, make_model
    with pytest.warns(UserWarning, match="outside the HIP kernel's envelope"):
        wide = _build("bojanet", 18)
    assert wide.backbone.native is False and sum(p.numel() for p in wide.parameters()) == 2 * 18 * 18 + 28 * 18 + 194
    rng = np.random.RandomState(3)
    amp, ph = 0.1 + 0.8 * rng.rand(2, 40, 1), 2 * np.pi * rng.rand(2, 40, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    p = np.concatenate([v.detach().numpy().reshape(-1) for v in wide.parameters()])
    xt = torch.from_numpy(x).requires_grad_(True)
    y = wide(xt)
    o, m = Oracle("f32"), make_model("bojanet", 18)
    yo, _ = o.forward(m, p, x)
    assert np.abs(y.detach().numpy() - yo).max() < 2e-5 * max(1.0, np.abs(yo).max())
    dy = rng.randn(2, 40, 2).astype(np.float32)
    y.backward(torch.from_numpy(dy))
    go, dxo = o.backward(m, p, x, dy, need_dx=True)
    g = np.concatenate([v.grad.numpy().reshape(-1) for v in wide.parameters()])
    assert np.abs(g - go).max() < 2e-4 * np.abs(go).max() and np.abs(xt.grad.numpy() - dxo).max() < 2e-4 * np.abs(dxo).max()
    with pytest.raises(Exception):                    # beyond 18 the reference's own phase re-rotation cannot be built (bojanet.py:41-53)
        _build("bojanet", 19)(torch.from_numpy(x))


def test_apnrru_is_native_and_constructs_like_the_reference():
    """apnrru left this module for csrc/apnrru_s16.hip (hidden <= 14); the seeded construction — including the reference's
    reset_parameters() that stops at a missing attribute and leaves the read-outs at their default draw — still reproduces the
    reference's state dict and RNG consumption; beyond the envelope the restatement serves it, and keeps computing what the reference does."""
    fx = Fixture("extra_apnrru_h8")
    net = _build("apnrru", 8)
    after = float(torch.rand(1))
    sd = net.state_dict()
    assert list(sd.keys()) == fx.keys("sd")
    for k in sd:
        assert np.array_equal(sd[k].numpy(), fx["sd/" + k]), k
    assert after == fx.meta["rng_after_init"]
    assert net.backbone.native is True and sum(p.numel() for p in net.parameters()) == fx.meta["n_param"] == 343 + 70 * 8
    with pytest.warns(UserWarning, match="envelope"):
        wide = _build("apnrru", 15)
    assert wide.backbone.native is False
    from opendpd_amd.backbones.extras import APNRRU
    ref = APNRRU(hidden_size=8)
    ref.load_state_dict({k[len("backbone."):]: torch.from_numpy(fx["sdu/" + k]) for k in fx.keys("sdu")})
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    y = ref(x)
    assert rel_err(y.detach().numpy(), fx["y"]) < TOL
    torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"])).backward()
    for k, p in ref.named_parameters():
        assert rel_err(p.grad.numpy(), fx["g/backbone." + k]) < 10 * TOL, k
    assert rel_err(x.grad.numpy(), fx["gx"]) < 10 * TOL


def test_mcldnn_is_native_and_constructs_like_the_reference():
    """mcldnn left this module for csrc/mcldnn.hip (<= 16 channels); the seeded construction — the constructor's own draw and the
    registry's second reset_parameters() — still reproduces the reference's state dict and RNG consumption; beyond the envelope the
    restatement serves it, with a warning, and keeps computing what the reference computes."""
    fx = Fixture("extra_mcldnn_h8")
    net = _build("mcldnn", 8)
    after = float(torch.rand(1))
    sd = net.state_dict()
    assert list(sd.keys()) == fx.keys("sd")
    for k in sd:
        assert np.array_equal(sd[k].numpy(), fx["sd/" + k]), k
    assert after == fx.meta["rng_after_init"]
    assert net.backbone.native is True and sum(p.numel() for p in net.parameters()) == fx.meta["n_param"] == 190 * 8 + 589
    with pytest.warns(UserWarning, match="envelope"):
        wide = _build("mcldnn", 20)
    assert wide.backbone.native is False and sum(p.numel() for p in wide.parameters()) == 190 * 20 + 589
    from opendpd_amd.backbones.extras import MCLDNN
    ref = MCLDNN(hidden_size=8)
    ref.load_state_dict({k[len("backbone."):]: torch.from_numpy(fx["sdu/" + k]) for k in fx.keys("sdu")})
    x = torch.from_numpy(fx["x"]).requires_grad_(True)
    y = ref(x)
    assert rel_err(y.detach().numpy(), fx["y"]) < TOL
    loss = torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"]))
    assert abs(float(loss.detach()) - fx.meta["loss"]) < 1e-6 * max(1.0, fx.meta["loss"])
    loss.backward()
    for k, p in ref.named_parameters():
        assert rel_err(p.grad.numpy(), fx["g/backbone." + k]) < 10 * TOL, k
    assert rel_err(x.grad.numpy(), fx["gx"]) < 10 * TOL
