"""GPU parity of the DeltaJANET kernels (the JAN instantiation of csrc/delta_s16.hip; reference backbones/deltajanet.py:11-274) against
vectors produced by the reference (tests/golden/deltajanet_h{15,22}.npz, extra_deltajanet_h10.npz) and against the CPU oracle on
ragged shapes at hidden sizes of both unit-tile classes: outputs, sparsity counters, parameter gradients, dL/dx, trajectory."""
import numpy as np
import pytest
import torch

from tests.golden_util import Fixture, rel_err

pytestmark = pytest.mark.gpu
FWD_TOL, GRAD_TOL = 2e-5, 3e-4


def _net(H, fx=None, prefix="sd", **kw):
    from opendpd_amd import CoreModel
    net = CoreModel(2, H, 1, "deltajanet", **kw)
    if fx is not None:
        net.load_state_dict({k: torch.from_numpy(fx[f"{prefix}/" + k]) for k in fx.keys(prefix)})
    return net.cuda()


@pytest.mark.parametrize("name", ["deltajanet_h15", "deltajanet_h22"])
def test_golden_forward_backward_and_counters(name):
    fx = Fixture(name)
    net = _net(fx.meta["hidden"], fx, thx=fx.meta["thx"], thh=fx.meta["thh"])
    assert net.backbone.native and sum(p.numel() for p in net.parameters()) == fx.meta["n_param"]
    net.backbone.set_debug(1)
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_(True)
    y = net(x)
    assert rel_err(y.detach().cpu().numpy(), fx["y"]) < FWD_TOL
    st = net.backbone.statistics
    got = np.array([st["num_dx_zeros"], st["num_dx_numel"], st["num_dh_zeros"], st["num_dh_numel"]])
    assert np.array_equal(got, fx["stats"]), (got, fx["stats"])        # thresholds are dropped by the wrapper: exact repeats only
    loss = torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"]).cuda())
    assert abs(loss.item() - fx["losses"][0]) < 1e-5 * max(1.0, fx["losses"][0])
    loss.backward()
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), fx["g/" + k]) < GRAD_TOL, k
    assert rel_err(x.grad.cpu().numpy(), fx["gx"]) < GRAD_TOL
    net.backbone.set_debug(1)
    with torch.no_grad():
        ya = net(torch.from_numpy(fx["xa"]).cuda())
    assert rel_err(ya.cpu().numpy(), fx["ya"]) < FWD_TOL
    st = net.backbone.statistics
    assert np.array_equal(np.array([st["num_dx_zeros"], st["num_dx_numel"], st["num_dh_zeros"], st["num_dh_numel"]]), fx["stats_a"])
    sp = net.backbone.get_temporal_sparsity()
    assert set(sp) == {"SP_T_DX", "SP_T_DH", "SP_T_DV"}


def test_second_reference_vector():
    fx = Fixture("extra_deltajanet_h10")
    net = _net(10, fx, "sdu", thx=0.01, thh=0.05)
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_(True)
    y = net(x)
    assert rel_err(y.detach().cpu().numpy(), fx["y"]) < FWD_TOL
    torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"]).cuda()).backward()
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), fx["g/" + k]) < GRAD_TOL, k
    assert rel_err(x.grad.cpu().numpy(), fx["gx"]) < GRAD_TOL


@pytest.mark.parametrize("H", [1, 7, 15, 16, 17, 22, 32])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (17, 32), (7, 33), (5, 200), (66, 63)])
def test_against_oracle_ragged(H, B, T):
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H * 100 + B + T)
    net = _net(H)
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    rng = np.random.RandomState(B * 17 + T)
    amp = 0.05 + 0.85 * rng.rand(B, T, 1)
    ph = 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    dy = rng.randn(B, T, 2).astype(np.float32)
    o = Oracle("f32")
    m = make_model("deltajanet", H)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    yo, so = o.forward(m, p, x)
    go, dxo = o.backward(m, p, x, dy)
    # weight gradients alone, then with dL/dx, then dL/dx of the frozen model
    net.backbone.set_debug(1)
    y = net(torch.from_numpy(x).cuda())
    y.backward(torch.from_numpy(dy).cuda())
    assert rel_err(y.detach().cpu().numpy(), yo) < FWD_TOL
    st = net.backbone.statistics
    assert st["num_dx_zeros"] == so[0] and st["num_dh_zeros"] == so[2] and st["num_dx_numel"] == so[1] and st["num_dh_numel"] == so[3]
    g = np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in net.parameters()])
    assert rel_err(g, go) < GRAD_TOL
    for q in net.parameters():
        q.grad = None
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    net(xt).backward(torch.from_numpy(dy).cuda())
    g = np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in net.parameters()])
    assert rel_err(g, go) < GRAD_TOL
    assert rel_err(xt.grad.cpu().numpy(), dxo) < GRAD_TOL
    for q in net.parameters():
        q.requires_grad_(False)
    xt2 = torch.from_numpy(x).cuda().requires_grad_(True)
    net(xt2).backward(torch.from_numpy(dy).cuda())
    assert rel_err(xt2.grad.cpu().numpy(), dxo) < GRAD_TOL


@pytest.mark.parametrize("name", ["deltajanet_h15", "deltajanet_h22"])
def test_train_steps_follow_reference(name):
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    fx = Fixture(name)
    net = _net(fx.meta["hidden"], fx)
    opt = FusedAdamW(net, lr=fx.meta["lr"])
    x = torch.from_numpy(fx["x"]).cuda()
    t = torch.from_numpy(fx["tgt"]).cuda()
    names = fx.keys("sd")
    for s in range(1, 4):
        loss = fused_train_step(opt, x, t, "l2", fx.meta["clip"])
        assert abs(loss.item() - fx["losses"][s - 1]) < 2e-5 * max(1.0, fx["losses"][s - 1])
        assert rel_err(net.backbone.flat_params().detach().cpu().numpy(), fx.flat(f"p{s}", names)) < 3e-5, s


def test_cascade_roles():
    """deltajanet as the DPD in front of a frozen DGRU PA and as the frozen PA behind a GRU DPD, against the oracle composition"""
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    o = Oracle("f32")
    rng = np.random.RandomState(0)
    x = (rng.uniform(0.05, 0.7, (9, 41, 2)) * rng.choice([-1.0, 1.0], (9, 41, 2))).astype(np.float32)
    for dpd_bb, dH, pa_bb, pH in (("deltajanet", 15, "dgru", 13), ("gru", 11, "deltajanet", 20)):
        torch.manual_seed(3)
        casc = CascadedModel(dpd_model=CoreModel(2, dH, 1, dpd_bb), pa_model=CoreModel(2, pH, 1, pa_bb))
        casc.freeze_pa_model()
        casc = casc.cuda()
        pd = torch.cat([q.detach().reshape(-1) for q in casc.dpd_model.parameters()]).cpu().numpy()
        pp = torch.cat([q.detach().reshape(-1) for q in casc.pa_model.parameters()]).cpu().numpy()
        md, mp = make_model(dpd_bb, dH), make_model(pa_bb, pH)
        u, _ = o.forward(md, pd, x)
        y, _ = o.forward(mp, pp, u)
        lo, dy = o.loss("l2", y, x)
        _, du = o.backward(mp, pp, u, dy)
        gd, _ = o.backward(md, pd, x, du, need_dx=False)
        opt = FusedAdamW(casc, lr=0.0, weight_decay=0.0)
        xt = torch.from_numpy(x).cuda()
        loss = fused_train_step(opt, xt, xt.clone(), "l2", 0.0)
        assert abs(loss.item() - lo) < 1e-5 * max(1.0, lo), (dpd_bb, pa_bb)
        assert rel_err(opt.grad[:-4].cpu().numpy(), gd) < GRAD_TOL, (dpd_bb, pa_bb)
