"""GPU parity of the quantisation-aware QGRU kernels (csrc/qgru_family.hip).  For 8-bit grids the HIP path must be
BIT-EXACT with the reference (train-mode float outputs, eval-mode outputs on the 2^-14 grid, before and after training);
for 16-bit grids (32-bit products, fp32 accumulation order matters) qgru is bit-exact as well and qgru_amp1 agrees to <= 2.5 LSB of the
2^-14 output grid on <= 8 % of the outputs (measured: 2.19 LSB, 6 %)."""
import numpy as np
import pytest
import torch

from tests.golden_util import Fixture, rel_err
from tests.test_oracle_golden import QAT, qat_param_names

pytestmark = pytest.mark.gpu


class _Proj:
    quant = True
    pretrained_model = ""


def _qmodel(fx, bb, bits, prefix="sd"):
    from opendpd_amd import CoreModel
    from opendpd_amd.quant import get_quant_model
    _Proj.n_bits_w = _Proj.n_bits_a = bits
    q = get_quant_model(_Proj, CoreModel(2, fx.meta["hidden"], 1, bb))
    q.load_state_dict({k: torch.from_numpy(fx[f"{prefix}/" + k]) for k in fx.keys(prefix)})
    return q.cuda()


@pytest.mark.parametrize("name,bb,bits", QAT)
def test_forward_train_eval_bit_exact(name, bb, bits):
    fx = Fixture(name)
    x = torch.from_numpy(fx["x"]).cuda()
    for prefix, ytr, yev in (("sd", "y", "y_eval"), ("sd3", "y_p3_train", "y_p3_eval")):
        q = _qmodel(fx, bb, bits, prefix)
        q.train()
        with torch.no_grad():
            yt = q(x).cpu().numpy()
        q.eval()
        with torch.no_grad():
            ye = q(x).cpu().numpy()
        if bits == 8:
            assert np.array_equal(yt, fx[ytr]), (prefix, np.abs(yt - fx[ytr]).max())
            assert np.array_equal(ye, fx[yev]), prefix
        else:
            # 16-bit grids: 32-bit products summed in fp32, so the summation order decides the last bit of a pre-activation; where that lands
            # within rounding reach of a 2^-14 grid boundary the quantised value moves by one grid step and the recurrence carries it on.
            # Measured (tools/probe_w16.py): qgru bit-exact; qgru_amp1 <= 1 LSB on 7 of 370 outputs at initialisation, <= 2.19 LSB on 22 of
            # 370 after three training steps.  Asserted: <= 2.5 LSB of the 2^-14 output grid, on at most 8 % of the outputs
            lsb = 2.0 ** -14
            for got, want in ((yt, fx[ytr]), (ye, fx[yev])):
                d = np.abs(got - want)
                assert d.max() <= 2.5 * lsb, (name, prefix, d.max() / lsb)
                assert (d > 0.5 * lsb).mean() <= 0.08, (name, prefix, (d > 0.5 * lsb).mean())
            if bb == "qgru":
                assert np.array_equal(yt, fx[ytr]) and np.array_equal(ye, fx[yev]), prefix


@pytest.mark.parametrize("name,bb,bits", QAT)
def test_gradients_and_trajectory(name, bb, bits):
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    fx = Fixture(name)
    q = _qmodel(fx, bb, bits)
    q.train()
    x = torch.from_numpy(fx["x"]).cuda()
    t = torch.from_numpy(fx["tgt"]).cuda()
    loss = torch.nn.functional.mse_loss(q(x), t)
    loss.backward()
    tol = 1e-5 if bits == 8 else 2e-3
    for k, p in q.named_parameters():
        if ("g/" + k) in fx:
            assert rel_err(p.grad.cpu().numpy(), fx["g/" + k]) < tol or np.abs(fx["g/" + k]).max() == 0, k
            if "scale" in k:
                assert float(p.grad.abs().max()) == 0.0
    names = qat_param_names(fx)
    opt = FusedAdamW(q, lr=fx.meta["lr"])
    for s in range(1, 4):
        l = fused_train_step(opt, x, t, "l2", fx.meta["clip"])
        assert abs(l.item() - fx["losses"][s - 1]) < (2e-6 if bits == 8 else 1e-4)
        got = np.concatenate([p.detach().cpu().numpy().reshape(-1) for p in q.parameters()])
        assert rel_err(got, fx.flat(f"p{s}", names)) < (2e-6 if bits == 8 else 1e-4), s


@pytest.mark.parametrize("bb", ["qgru", "qgru_amp1"])
@pytest.mark.parametrize("H,B,T", [(10, 3, 5), (16, 7, 33), (6, 66, 63), (13, 5, 200)])
def test_w8a8_matches_oracle_bitwise_on_ragged_sizes(bb, H, B, T):
    from opendpd_amd import CoreModel
    from opendpd_amd.quant import get_quant_model
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H + B + T)
    _Proj.n_bits_w = _Proj.n_bits_a = 8
    q = get_quant_model(_Proj, CoreModel(2, H, 1, bb)).cuda()
    rng = np.random.RandomState(B + T)
    amp = 0.05 + 0.85 * rng.rand(B, T, 1)
    ph = 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    dy = rng.randn(B, T, 2).astype(np.float32)
    q.train()
    y = q(torch.from_numpy(x).cuda())
    y.backward(torch.from_numpy(dy).cuda())
    o = Oracle("f32")
    m = make_model(bb, H, bits_w=8, bits_a=8)
    p = np.concatenate([v.detach().cpu().numpy().reshape(-1) for v in q.parameters()])
    yo = o.qat_forward(m, p, x)
    assert np.array_equal(y.detach().cpu().numpy(), yo)
    go, _ = o.qat_backward(m, p, x, dy, need_dx=False)
    g = np.concatenate([(v.grad if v.grad is not None else torch.zeros_like(v)).cpu().numpy().reshape(-1) for v in q.parameters()])
    assert rel_err(g, go) < 2e-5


@pytest.mark.parametrize("bb", ["qgru", "qgru_amp1"])
@pytest.mark.parametrize("H,B,T", [(10, 3, 5), (16, 7, 33), (6, 66, 63), (13, 5, 200)])
def test_w8a8_input_gradient_matches_oracle(bb, H, B, T):
    """dL/dx of the quantised cell (straight-through over the x2h activation quantiser, quantised W_x^T, feature Jacobian):
    together with the weight gradients and alone (frozen model = PA of a cascade)."""
    from opendpd_amd import CoreModel
    from opendpd_amd.quant import get_quant_model
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H + B + T)
    _Proj.n_bits_w = _Proj.n_bits_a = 8
    q = get_quant_model(_Proj, CoreModel(2, H, 1, bb)).cuda()
    rng = np.random.RandomState(B + T)
    amp = 0.05 + 0.85 * rng.rand(B, T, 1)
    ph = 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    dy = rng.randn(B, T, 2).astype(np.float32)
    q.train()
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    q(xt).backward(torch.from_numpy(dy).cuda())
    o = Oracle("f32")
    m = make_model(bb, H, bits_w=8, bits_a=8)
    p = np.concatenate([v.detach().cpu().numpy().reshape(-1) for v in q.parameters()])
    go, dxo = o.qat_backward(m, p, x, dy, need_dx=True)
    g = np.concatenate([(v.grad if v.grad is not None else torch.zeros_like(v)).cpu().numpy().reshape(-1) for v in q.parameters()])
    assert rel_err(g, go) < 2e-5
    assert rel_err(xt.grad.cpu().numpy(), dxo) < 2e-5
    for v in q.parameters():
        v.requires_grad_(False)
    xt2 = torch.from_numpy(x).cuda().requires_grad_(True)
    q(xt2).backward(torch.from_numpy(dy).cuda())
    assert rel_err(xt2.grad.cpu().numpy(), dxo) < 2e-5


def test_checkpoint_buffers_and_mode_flag_follow_the_reference():
    """(1) pow2_scale / decimal_num / integer_num are side effects of the reference's quantiser forwards: after training steps plus a
    train-mode and an eval-mode forward every exercised quantiser holds its refreshed values, the never-used x2h / h2h
    out_quantizers keep (0, 1, 14) — the whole state dict equals the reference's `sd3` fixture, buffers included.
    (2) an evaluation pass must not leave ODPD_FLAG_EVAL behind for the next direct-ABI train step: a model that was evaluated
    between two steps ends up bit-identical with one that was not."""
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    fx = Fixture("quant_qgru_h10_w8a8")
    x, t = torch.from_numpy(fx["x"]).cuda(), torch.from_numpy(fx["tgt"]).cuda()
    nets = []
    for evaluate_between in (False, True):
        q = _qmodel(fx, "qgru", 8)
        opt = FusedAdamW(q, lr=fx.meta["lr"])
        q.train()
        for s in range(3):
            fused_train_step(opt, x, t, "l2", fx.meta["clip"])
            if evaluate_between:
                q.eval()
                with torch.no_grad():
                    q(x)
                q.train()
        nets.append(q)
    a, b = nets[0].backbone.flat_params(), nets[1].backbone.flat_params()
    assert torch.equal(a, b)
    q = nets[0]
    sd_train_only = {k: v.cpu().numpy() for k, v in q.state_dict().items()}
    assert sd_train_only["backbone.fc_out.out_quantizer.decimal_num"][0] == 1.0          # eval has not run on this one yet
    assert sd_train_only["backbone.fc_out.weight_quantizer.decimal_num"][0] == 6.0
    q.eval()
    with torch.no_grad():
        q(x)
    sd = q.state_dict()
    for k in fx.keys("sd3"):
        tol = 0 if ("_num" in k or "pow2" in k or "n_bits" in k) else 3e-5
        assert np.abs(sd[k].cpu().numpy() - fx["sd3/" + k]).max() <= tol * max(1.0, np.abs(fx["sd3/" + k]).max()), k
