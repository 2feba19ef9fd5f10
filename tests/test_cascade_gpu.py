"""train_dpd path: CascadedModel(DPD, frozen PA) — reference models.py:163-176, steps/train_dpd.py:60-63.
Checks the autograd path and the fused five-launch step against the reference's cascade fixtures."""
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
