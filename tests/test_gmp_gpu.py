"""GPU parity of the GMP kernels (csrc/gmp.hip; reference backbones/gmp.py:5-50) against vectors produced by the reference
and against the CPU oracle on ragged / long / chunked sizes, plus the split train step trajectory."""
import numpy as np
import pytest
import torch

from tests.golden_util import Fixture, rel_err

pytestmark = pytest.mark.gpu
FWD_TOL, GRAD_TOL = 2e-5, 1e-4


def _net(weight=None):
    from opendpd_amd import CoreModel
    net = CoreModel(2, 11, 1, "gmp")
    if weight is not None:
        net.load_state_dict({"backbone.Weight": torch.from_numpy(weight)})
    return net.cuda()


def test_golden_forward_backward():
    fx = Fixture("gmp_m11")
    net = _net(fx["sd/backbone.Weight"])
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_(True)
    y = net(x)
    assert rel_err(y.detach().cpu().numpy(), fx["y"]) < FWD_TOL
    loss = torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"]).cuda())
    assert abs(loss.item() - fx["losses"][0]) < 1e-5 * max(1.0, fx["losses"][0])
    loss.backward()
    assert rel_err(net.backbone.Weight.grad.cpu().numpy(), fx["g/backbone.Weight"]) < GRAD_TOL
    assert rel_err(x.grad.cpu().numpy(), fx["gx"]) < GRAD_TOL
    with torch.no_grad():
        ya = net(torch.from_numpy(fx["xa"]).cuda())          # config-shaped frames (8, 200, 2) of APA_200MHz
    assert rel_err(ya.cpu().numpy(), fx["ya"]) < FWD_TOL


def test_second_reference_vector():
    """the vectors that pinned the former torch restatement (oracle/gen_golden_extras.py)"""
    fx = Fixture("extra_gmp_h8")
    net = _net(fx["sdu/backbone.Weight"])
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_(True)
    y = net(x)
    assert rel_err(y.detach().cpu().numpy(), fx["y"]) < FWD_TOL
    torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"]).cuda()).backward()
    assert rel_err(net.backbone.Weight.grad.cpu().numpy(), fx["g/backbone.Weight"]) < GRAD_TOL
    assert rel_err(x.grad.cpu().numpy(), fx["gx"]) < GRAD_TOL


# frames shorter than the memory, ragged tails of the 256-lane passes, several frames per LDS region, records longer than one
# 512-sample chunk (halo on both sides), a batch large enough for several regions per workgroup
@pytest.mark.parametrize("B,T", [(1, 1), (1, 3), (3, 5), (2, 11), (4, 32), (7, 33), (5, 200), (66, 63), (2, 700), (1, 1500), (3, 513),
                                 (700, 50), (1100, 20)])
def test_against_oracle(B, T):
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(B * 7 + T)
    net = _net()
    with torch.no_grad():
        net.backbone.Weight.mul_(4.0)                          # make the envelope terms count
    rng = np.random.RandomState(B * 11 + T)
    amp = 0.05 + 0.85 * rng.rand(B, T, 1)
    ph = 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    if T > 4:
        x[0, 2] = 0.0                                           # |x| = 0: the envelope derivative is taken as 0 there (torch.abs)
    dy = rng.randn(B, T, 2).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    y = net(xt)
    y.backward(torch.from_numpy(dy).cuda())
    o = Oracle("f32")
    m = make_model("gmp", 11)
    p = net.backbone.Weight.detach().cpu().numpy().reshape(-1)
    yo, _ = o.forward(m, p, x)
    go, dxo = o.backward(m, p, x, dy)
    assert rel_err(y.detach().cpu().numpy(), yo) < FWD_TOL
    assert rel_err(net.backbone.Weight.grad.cpu().numpy().reshape(-1), go) < GRAD_TOL
    assert rel_err(xt.grad.cpu().numpy(), dxo) < GRAD_TOL


def test_frozen_gives_dx_only():
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(3)
    net = _net()
    for p in net.parameters():
        p.requires_grad_(False)
    rng = np.random.RandomState(3)
    x = (rng.uniform(0.05, 0.9, (6, 77, 2)) * rng.choice([-1.0, 1.0], (6, 77, 2))).astype(np.float32)
    dy = rng.randn(6, 77, 2).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    net(xt).backward(torch.from_numpy(dy).cuda())
    _, dxo = Oracle("f32").backward(make_model("gmp", 11), net.backbone.Weight.cpu().numpy().reshape(-1), x, dy)
    assert rel_err(xt.grad.cpu().numpy(), dxo) < GRAD_TOL
    assert net.backbone.Weight.grad is None


def test_train_steps_follow_reference():
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    fx = Fixture("gmp_m11")
    net = _net(fx["sd/backbone.Weight"])
    opt = FusedAdamW(net, lr=fx.meta["lr"])
    x = torch.from_numpy(fx["x"]).cuda()
    t = torch.from_numpy(fx["tgt"]).cuda()
    for s in range(1, 4):
        loss = fused_train_step(opt, x, t, "l2", fx.meta["clip"])
        assert abs(loss.item() - fx["losses"][s - 1]) < 2e-5 * max(1.0, fx["losses"][s - 1])
        assert rel_err(net.backbone.Weight.detach().cpu().numpy(), fx[f"p{s}/backbone.Weight"]) < 3e-5, s


def test_other_configurations_are_refused_loudly():
    from opendpd_amd import backbones as B
    with pytest.raises(NotImplementedError):
        B.GMP(memory_length=7)


@pytest.mark.parametrize("B,T", [(5, 37), (256, 200), (3, 513), (2, 1300), (700, 50), (1, 1)])
def test_fused_step_equals_split_kernels(B, T):
    """single-launch train step (x, target -> forward -> loss and dy in LDS -> MFMA weight gradient; L2 and L1) == autograd through
    the split kernels (oracle-checked above); records longer than one chunk walk the 10 samples after each chunk twice"""
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    torch.manual_seed(1)
    net = _net()
    with torch.no_grad():
        net.backbone.Weight.mul_(4.0)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.6
    x = x + 0.05 * torch.sign(x)
    t = torch.randn(B, T, 2, device="cuda", generator=g) * 0.3
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    assert opt.has_fused(B, T) and opt.train_workspace(B, T, x.device) is None
    for kind, fn in (("l2", torch.nn.functional.mse_loss), ("l1", torch.nn.functional.l1_loss)):
        net.backbone.Weight.grad = None
        loss = fn(net(x), t)
        loss.backward()
        gref = net.backbone.Weight.grad.reshape(-1).cpu().numpy()
        lf = fused_train_step(opt, x, t, kind, 0.0)
        assert abs(lf.item() - loss.item()) < 1e-5 * max(1.0, loss.item()), kind
        assert rel_err(opt.grad[:-4].cpu().numpy(), gref) < 2e-5, kind


def test_native_epoch_loop_reads_frames_in_place():
    from tests import test_e2e_gpu as e2e
    e2e.test_native_epoch_loop_equals_per_step_loop("gmp", 11, 50, 64)
    e2e.test_native_epoch_loop_equals_per_step_loop("gmp", 11, 200, 256)
