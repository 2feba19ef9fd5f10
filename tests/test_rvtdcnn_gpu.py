"""GPU parity of the RVTDCNN kernels (csrc/rvtdcnn.hip; reference backbones/rvtdcnn.py:9-62) against vectors produced by the
reference (tests/golden/rvtdcnn_h{25,6}.npz, extra_rvtdcnn_h6.npz) and against the CPU oracle on ragged / long / chunked sizes at
every hidden size class, plus the train-step trajectory and the single-launch train step."""
import numpy as np
import pytest
import torch

from tests.golden_util import Fixture, rel_err

pytestmark = pytest.mark.gpu
FWD_TOL, GRAD_TOL = 2e-5, 2e-4


def _net(H, fx=None, prefix="sd"):
    from opendpd_amd import CoreModel
    net = CoreModel(2, H, 1, "rvtdcnn")
    if fx is not None:
        net.load_state_dict({k: torch.from_numpy(fx[f"{prefix}/" + k]) for k in fx.keys(prefix)})
    return net.cuda()


def _flat_grad(net):
    return torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()


@pytest.mark.parametrize("name", ["rvtdcnn_h25", "rvtdcnn_h6"])
def test_golden_forward_backward(name):
    fx = Fixture(name)
    net = _net(fx.meta["hidden"], fx)
    assert net.backbone.native and sum(p.numel() for p in net.parameters()) == fx.meta["n_param"]
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_(True)
    y = net(x)
    assert rel_err(y.detach().cpu().numpy(), fx["y"]) < FWD_TOL
    loss = torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"]).cuda())
    assert abs(loss.item() - fx["losses"][0]) < 1e-5 * max(1.0, fx["losses"][0])
    loss.backward()
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), fx["g/" + k]) < GRAD_TOL, k
    assert rel_err(x.grad.cpu().numpy(), fx["gx"]) < GRAD_TOL
    with torch.no_grad():
        ya = net(torch.from_numpy(fx["xa"]).cuda())          # config-shaped frames (8, 200, 2) of APA_200MHz
    assert rel_err(ya.cpu().numpy(), fx["ya"]) < FWD_TOL


def test_second_reference_vector():
    """the vectors that pinned the former torch restatement (oracle/gen_golden_extras.py)"""
    fx = Fixture("extra_rvtdcnn_h6")
    net = _net(6, fx, "sdu")
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_(True)
    y = net(x)
    assert rel_err(y.detach().cpu().numpy(), fx["y"]) < FWD_TOL
    torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"]).cuda()).backward()
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), fx["g/" + k]) < GRAD_TOL, k
    assert rel_err(x.grad.cpu().numpy(), fx["gx"]) < GRAD_TOL


# the shortest frame the circular window admits, ragged tails of the 256-thread passes, several frames per pass of the dL/dx layout,
# frames longer than one 253-sample chunk (halo wraps into the next chunk / around the frame), hidden sizes of both tile classes
_SHAPES = [(1, 3), (3, 5), (2, 11), (7, 33), (5, 200), (66, 63), (2, 700), (1, 1500), (3, 253), (3, 254), (700, 50)]
_CORE = [(3, 5), (5, 200), (2, 700), (66, 63)]        # the full shape sweep runs at two hidden sizes, four shapes at the others


@pytest.mark.parametrize("H,B,T", [(H, B, T) for H in (1, 6, 16, 17, 25, 32) for (B, T) in (_SHAPES if H in (6, 25) else _CORE)])
def test_against_oracle(H, B, T):
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(B * 7 + T + H)
    net = _net(H)
    rng = np.random.RandomState(B * 11 + T)
    amp = 0.05 + 0.85 * rng.rand(B, T, 1)
    ph = 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    dy = rng.randn(B, T, 2).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    y = net(xt)
    y.backward(torch.from_numpy(dy).cuda())
    o = Oracle("f32")
    m = make_model("rvtdcnn", H)
    p = torch.cat([q.detach().reshape(-1) for q in net.parameters()]).cpu().numpy()
    assert o.param_count(m) == p.size
    yo, _ = o.forward(m, p, x)
    go, dxo = o.backward(m, p, x, dy)
    assert rel_err(y.detach().cpu().numpy(), yo) < FWD_TOL
    assert rel_err(_flat_grad(net), go) < GRAD_TOL
    assert rel_err(xt.grad.cpu().numpy(), dxo) < GRAD_TOL
    # weight gradients alone (the flat thread layout) and dL/dx alone (frozen model)
    for q in net.parameters():
        q.grad = None
    net(torch.from_numpy(x).cuda()).backward(torch.from_numpy(dy).cuda())
    assert rel_err(_flat_grad(net), go) < GRAD_TOL
    for q in net.parameters():
        q.requires_grad_(False)
    xt2 = torch.from_numpy(x).cuda().requires_grad_(True)
    net(xt2).backward(torch.from_numpy(dy).cuda())
    assert rel_err(xt2.grad.cpu().numpy(), dxo) < GRAD_TOL


def test_small_amplitudes_keep_relative_accuracy():
    """every feature scales with the signal amplitude: the conv pre-activations of a quiet frame are tiny and tanh must keep its
    relative accuracy there"""
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(2)
    net = _net(25)
    rng = np.random.RandomState(2)
    o = Oracle("f64")
    m = make_model("rvtdcnn", 25)
    p = torch.cat([q.detach().reshape(-1) for q in net.parameters()]).cpu().numpy()
    for scale in (1.0, 0.1, 0.01):
        x = (scale * rng.uniform(0.05, 0.9, (4, 64, 2)) * rng.choice([-1.0, 1.0], (4, 64, 2))).astype(np.float32)
        with torch.no_grad():
            y = net(torch.from_numpy(x).cuda()).cpu().numpy()
        yo, _ = o.forward(m, p, x)
        assert rel_err(y, yo) < 2e-6, scale


@pytest.mark.parametrize("name", ["rvtdcnn_h25", "rvtdcnn_h6"])
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
    assert rel_err(opt.exp_avg.cpu().numpy(), fx.flat("m3", names)) < 1e-3
    assert rel_err(opt.exp_avg_sq.cpu().numpy(), fx.flat("v3", names)) < 1e-3


@pytest.mark.parametrize("H", [6, 25])
@pytest.mark.parametrize("B,T", [(5, 37), (256, 200), (3, 513), (700, 50), (1, 3)])
def test_fused_step_equals_split_kernels(H, B, T):
    """single-launch train step (x, target -> forward -> loss -> weight gradients; L2 and L1) == autograd through the split kernels"""
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    torch.manual_seed(1)
    net = _net(H)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.6
    x = x + 0.05 * torch.sign(x)
    t = torch.randn(B, T, 2, device="cuda", generator=g) * 0.3
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    assert opt.has_fused(B, T) and opt.train_workspace(B, T, x.device) is None
    for kind, fn in (("l2", torch.nn.functional.mse_loss), ("l1", torch.nn.functional.l1_loss)):
        for p in net.parameters():
            p.grad = None
        loss = fn(net(x), t)
        loss.backward()
        gref = _flat_grad(net)
        lf = fused_train_step(opt, x, t, kind, 0.0)
        assert abs(lf.item() - loss.item()) < 1e-5 * max(1.0, loss.item()), kind
        assert rel_err(opt.grad[:-4].cpu().numpy(), gref) < 2e-5, kind


def test_fused_step_is_bit_repeatable():
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    g = torch.Generator(device="cuda").manual_seed(9)
    x = (torch.rand(3000, 50, 2, device="cuda", generator=g) - 0.5) * 1.6
    x = x + 0.05 * torch.sign(x)
    t = torch.randn(3000, 50, 2, device="cuda", generator=g) * 0.3
    res = []
    for _ in range(2):
        torch.manual_seed(4)
        net = _net(25)
        opt = FusedAdamW(net, lr=1e-3)
        for _ in range(3):
            fused_train_step(opt, x, t, "l2", 200.0)
        res.append(net.backbone.flat_params().clone())
    assert torch.equal(res[0], res[1])


def test_native_epoch_loop_reads_frames_in_place():
    from tests import test_e2e_gpu as e2e
    e2e.test_native_epoch_loop_equals_per_step_loop("rvtdcnn", 25, 50, 64)
    e2e.test_native_epoch_loop_equals_per_step_loop("rvtdcnn", 6, 200, 256)


def test_cascade_roles():
    """rvtdcnn as the DPD in front of a frozen DGRU PA, and as the frozen PA behind a GRU DPD: loss and DPD gradient of the train_dpd
    step against the oracle composition"""
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    o = Oracle("f32")
    rng = np.random.RandomState(0)
    x = (rng.uniform(0.05, 0.7, (9, 41, 2)) * rng.choice([-1.0, 1.0], (9, 41, 2))).astype(np.float32)
    for dpd_bb, dH, pa_bb, pH in (("rvtdcnn", 25, "dgru", 13), ("gru", 11, "rvtdcnn", 6)):
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
