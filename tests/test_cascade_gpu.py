"""train_dpd path: CascadedModel(DPD, frozen PA) — reference models.py:163-176, steps/train_dpd.py:60-63.
Checks the autograd path and the fused cascade step against the reference's cascade fixtures."""
import numpy as np
import pytest
import torch

from tests.golden_util import Fixture, rel_err

pytestmark = pytest.mark.gpu

CASES = [("cascade_gru11_gru11", "gru", "gru"), ("cascade_dgru13_dgru23", "dgru", "dgru"),
         ("cascade_tres15_dgru23", "deltagru_tcnskip", "dgru")]


def _cascade(fx, dpd_bb, pa_bb):
    from opendpd_amd import CascadedModel, CoreModel
    m = fx.meta
    dpd = CoreModel(2, m["dpd_hidden"], 1, dpd_bb, thx=m["thx"], thh=m["thh"])
    pa = CoreModel(2, m["pa_hidden"], 1, pa_bb)
    net = CascadedModel(dpd_model=dpd, pa_model=pa)
    net.load_state_dict({k: torch.from_numpy(fx["sd/" + k]) for k in fx.keys("sd")})
    net.freeze_pa_model()
    return net.cuda()


def _supported(bb):
    from opendpd_amd.models import CoreModel
    try:
        CoreModel(2, 8, 1, bb)
        return True
    except NotImplementedError:
        return False


@pytest.mark.parametrize("name,dpd_bb,pa_bb", CASES)
def test_cascade_autograd_matches_reference(name, dpd_bb, pa_bb):
    if not _supported(dpd_bb):
        pytest.skip(f"{dpd_bb} kernel not built yet")
    fx = Fixture(name)
    net = _cascade(fx, dpd_bb, pa_bb)
    x = torch.from_numpy(fx["x"]).cuda()
    y = net(x)
    assert rel_err(y.detach().cpu().numpy(), fx["y"]) < 2e-5
    loss = torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"]).cuda())
    assert abs(loss.item() - fx["losses"][0]) < 1e-5 * max(1.0, fx["losses"][0])
    loss.backward()
    assert all(p.grad is None for p in net.pa_model.parameters())
    for k, p in net.dpd_model.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), fx["g/dpd_model." + k]) < 3e-4, k


@pytest.mark.parametrize("name,dpd_bb,pa_bb", CASES)
def test_cascade_fused_steps_follow_reference(name, dpd_bb, pa_bb):
    if not _supported(dpd_bb):
        pytest.skip(f"{dpd_bb} kernel not built yet")
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    fx = Fixture(name)
    net = _cascade(fx, dpd_bb, pa_bb)
    opt = FusedAdamW(net, lr=fx.meta["lr"])
    x = torch.from_numpy(fx["x"]).cuda()
    t = torch.from_numpy(fx["tgt"]).cuda()
    names = [k for k in fx.keys("sd") if k.startswith("dpd_model.")]
    pa_before = net.pa_model.backbone.flat_params().clone()
    for s in range(1, 4):
        loss = fused_train_step(opt, x, t, "l2", fx.meta["clip"])
        assert abs(loss.item() - fx["losses"][s - 1]) < 2e-5 * max(1.0, fx["losses"][s - 1])
        got = np.concatenate([p.detach().cpu().numpy().reshape(-1) for p in net.dpd_model.parameters()])
        assert rel_err(got, fx.flat(f"p{s}", names)) < 3e-5, s
    assert torch.equal(pa_before, net.pa_model.backbone.flat_params())   # PA untouched


@pytest.mark.parametrize("pa_bb,pa_h", [("lstm", 14), ("vdlstm", 13), ("pgjanet", 11), ("tcnn", 35), ("dgru", 23), ("qgru_amp1", 10),
                                         ("deltagru", 15), ("deltagru_tcnskip", 12), ("gmp", 11)])
@pytest.mark.parametrize("dpd_bb,dpd_h", [("dgru", 9), ("deltagru_tcnskip", 15), ("gmp", 11)])
def test_cascade_with_every_pa_backbone_against_oracle(pa_bb, pa_h, dpd_bb, dpd_h):
    """Every float backbone with dL/dx can be the frozen PA of train_dpd: DPD gradient of the cascade step
    == oracle composition (DPD fwd, PA fwd, MSE, PA backward for dL/du only, DPD backward)."""
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(pa_h * 7 + dpd_h)
    B, T = 6, 70
    dpd = CoreModel(2, dpd_h, 1, dpd_bb)
    pa = CoreModel(2, pa_h, 1, pa_bb)
    net = CascadedModel(dpd_model=dpd, pa_model=pa)
    net.freeze_pa_model()
    net = net.cuda()
    rng = np.random.RandomState(pa_h)
    x = (rng.uniform(0.1, 0.8, (B, T, 2)) * rng.choice([-1.0, 1.0], (B, T, 2))).astype(np.float32)
    t = (0.5 * rng.randn(B, T, 2)).astype(np.float32)
    o = Oracle("f32")
    md, mp = make_model(dpd_bb, dpd_h), make_model(pa_bb, pa_h)
    pd = dpd.backbone.flat_params().detach().cpu().numpy().copy()
    pp = pa.backbone.flat_params().detach().cpu().numpy().copy()
    u, _ = o.forward(md, pd, x)
    y, _ = o.forward(mp, pp, u)
    lo, dy = o.loss("l2", y, t)
    _, du = o.backward(mp, pp, u, dy)
    gd, _ = o.backward(md, pd, x, du, need_dx=False)
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    lg = fused_train_step(opt, torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), "l2", 0.0)
    assert abs(lg.item() - lo) < 2e-5 * max(1.0, lo)
    assert rel_err(opt.grad[:-4].cpu().numpy(), gd) < 3e-4
    assert torch.equal(pa.backbone.flat_params().cpu(), torch.from_numpy(pp))


@pytest.fixture
def force_s16():
    from opendpd_amd import _lib
    lib = _lib.load()
    assert lib.odpd_set_tuning(b"s16_min_batch", 0) == 0
    yield
    lib.odpd_set_tuning(b"s16_min_batch", -1)


@pytest.mark.parametrize("pa_bb,pa_h", [("gru", 11), ("dgru", 13), ("dgru", 9), ("qgru", 10), ("qgru_amp1", 16), ("dgru", 23), ("gru", 30),
                                         ("qgru", 20), ("dgru", 28), ("dgru", 29), ("dgru", 31), ("dgru", 32), ("gru", 32), ("qgru_amp1", 29)])
@pytest.mark.parametrize("dpd_bb,dpd_h", [("dgru", 9), ("deltagru_tcnskip", 15)])
@pytest.mark.parametrize("loss", ["l2", "l1"])
def test_frozen_pa_single_launch_step_against_oracle(force_s16, pa_bb, pa_h, dpd_bb, dpd_h, loss):
    """GRU-family PA on the 16-sequences-per-wave kernels: forward + loss + dL/du of the frozen PA run as ONE launch
    (odpd_frozen_loss_dx: hidden <= 16 incl. the K-packed variant, hidden 17..32 with 2- and 4-chunk last tiles — hidden 29..32 are the
    sizes whose fourth K-chunk holds real units: an eight-wave build of that instantiation spilled and computed 1e-2-wrong results in
    r02, caught by tools/cascade_sweep.py); DPD gradient and loss of the cascade step == oracle composition."""
    from opendpd_amd import CascadedModel, CoreModel, _lib
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    import ctypes as C
    torch.manual_seed(pa_h * 7 + dpd_h)
    B, T = (37, 70) if pa_h % 2 else (16 * 9 + 5, 21)       # one loss row / several (10 sequence groups: 3 four-wave workgroups)
    dpd, pa = CoreModel(2, dpd_h, 1, dpd_bb), CoreModel(2, pa_h, 1, pa_bb)
    net = CascadedModel(dpd_model=dpd, pa_model=pa)
    net.freeze_pa_model()
    net = net.cuda()
    assert _lib.load().odpd_frozen_loss_rows(C.byref(pa.backbone.desc), B, T) > 0
    rng = np.random.RandomState(pa_h)
    x = (rng.uniform(0.1, 0.8, (B, T, 2)) * rng.choice([-1.0, 1.0], (B, T, 2))).astype(np.float32)
    t = (0.5 * rng.randn(B, T, 2)).astype(np.float32)
    o = Oracle("f32")
    md, mp = make_model(dpd_bb, dpd_h), make_model(pa_bb, pa_h)
    pd = dpd.backbone.flat_params().detach().cpu().numpy().copy()
    pp = pa.backbone.flat_params().detach().cpu().numpy().copy()
    u, _ = o.forward(md, pd, x)
    y, _ = o.forward(mp, pp, u)
    lo, dy = o.loss(loss, y, t)
    _, du = o.backward(mp, pp, u, dy)
    gd, _ = o.backward(md, pd, x, du, need_dx=False)
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    lg = fused_train_step(opt, torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), loss, 0.0)
    assert abs(lg.item() - lo) < 2e-5 * max(1.0, lo)
    assert rel_err(opt.grad[:-4].cpu().numpy(), gd) < (3e-4 if loss == "l2" else 2e-3)
    assert torch.equal(pa.backbone.flat_params().cpu(), torch.from_numpy(pp))
