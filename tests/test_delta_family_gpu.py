"""GPU parity of the delta-network GRU kernels (deltagru, deltagru_tcnskip = TRes-DeltaGRU): reference golden
vectors (outputs, parameter gradients, exact sparsity counters), CPU oracle on ragged sizes, train trajectory.

Note on tolerances: the delta thresholds make the forward map discontinuous (a |dh| within rounding of th_h can
flip a mask).  Inputs here are the same the oracle was pinned on; the counters are compared exactly."""
import numpy as np
import pytest
import torch

from tests.golden_util import Fixture, rel_err

pytestmark = pytest.mark.gpu
FWD_TOL, GRAD_TOL = 2e-5, 3e-4
CASES = [("deltagru_h15_dense", "deltagru"), ("deltagru_h15_th", "deltagru"), ("tres_h15_dense", "deltagru_tcnskip"),
         ("tres_h15_th", "deltagru_tcnskip"), ("deltagru_h24_th", "deltagru"), ("tres_h30_th", "deltagru_tcnskip")]


def _model(fx, bb):
    from opendpd_amd import CoreModel
    net = CoreModel(2, fx.meta["hidden"], 1, bb, thx=fx.meta["thx"], thh=fx.meta["thh"])
    net.load_state_dict({k: torch.from_numpy(fx["sd/" + k]) for k in fx.keys("sd")})
    return net.cuda()


@pytest.mark.parametrize("name,bb", CASES)
def test_golden_forward_backward_and_counters(name, bb):
    fx = Fixture(name)
    net = _model(fx, bb)
    net.backbone.set_debug(1)
    x = torch.from_numpy(fx["x"]).cuda()
    y = net(x)
    assert rel_err(y.detach().cpu().numpy(), fx["y"]) < FWD_TOL
    st = net.backbone.statistics
    got = np.array([st["num_dx_zeros"], st["num_dx_numel"], st["num_dh_zeros"], st["num_dh_numel"]])
    assert np.array_equal(got, fx["stats"]), (got, fx["stats"])
    loss = torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"]).cuda())
    loss.backward()
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), fx["g/" + k]) < GRAD_TOL, k
    net.backbone.set_debug(1)
    with torch.no_grad():
        ya = net(torch.from_numpy(fx["xa"]).cuda())
    assert rel_err(ya.cpu().numpy(), fx["ya"]) < FWD_TOL
    st = net.backbone.statistics
    got = np.array([st["num_dx_zeros"], st["num_dx_numel"], st["num_dh_zeros"], st["num_dh_numel"]])
    assert np.array_equal(got, fx["stats_a"]), (got, fx["stats_a"])
    sp = net.backbone.get_temporal_sparsity()
    assert abs(sp["SP_T_DH"] - fx["stats_a"][2] / fx["stats_a"][3]) < 1e-12 and "HW_PARAM" in sp


@pytest.mark.parametrize("bb,H,thx,thh", [("deltagru", 15, 0.0, 0.0), ("deltagru", 15, 0.01, 0.05), ("deltagru", 8, 0.02, 0.1),
                                          ("deltagru_tcnskip", 15, 0.01, 0.05), ("deltagru_tcnskip", 16, 0.0, 0.0),
                                          ("deltagru_tcnskip", 9, 0.05, 0.02)])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (4, 32), (7, 33), (5, 200), (66, 63)])
def test_against_oracle_ragged(bb, H, thx, thh, B, T):
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H * 100 + B + T)
    net = CoreModel(2, H, 1, bb, thx=thx, thh=thh).cuda()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    rng = np.random.RandomState(B * 13 + T)
    amp = 0.05 + 0.85 * rng.rand(B, T, 1)
    ph = 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    dy = rng.randn(B, T, 2).astype(np.float32)
    net.backbone.set_debug(1)
    y = net(torch.from_numpy(x).cuda())
    y.backward(torch.from_numpy(dy).cuda())
    o = Oracle("f32")
    m = make_model(bb, H, thx, thh)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    yo, so = o.forward(m, p, x)
    go, _ = o.backward(m, p, x, dy, need_dx=False)
    g = np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in net.parameters()])
    st = net.backbone.statistics
    # a rounding-level difference can flip a threshold decision; allow a handful of flips on the counters
    assert abs(st["num_dx_zeros"] - so[0]) <= 2 and abs(st["num_dh_zeros"] - so[2]) <= 2
    assert st["num_dx_numel"] == so[1] and st["num_dh_numel"] == so[3]
    exact = st["num_dx_zeros"] == so[0] and st["num_dh_zeros"] == so[2]
    tol_f, tol_g = (FWD_TOL, GRAD_TOL) if exact else (5e-3, 5e-2)
    assert rel_err(y.detach().cpu().numpy(), yo) < tol_f
    assert rel_err(g, go) < tol_g


@pytest.mark.parametrize("name,bb", [("deltagru_h15_th", "deltagru"), ("tres_h15_th", "deltagru_tcnskip"),
                                     ("deltagru_h24_th", "deltagru"), ("tres_h30_th", "deltagru_tcnskip")])
def test_train_steps_follow_reference(name, bb):
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    fx = Fixture(name)
    net = _model(fx, bb)
    opt = FusedAdamW(net, lr=fx.meta["lr"])
    x = torch.from_numpy(fx["x"]).cuda()
    t = torch.from_numpy(fx["tgt"]).cuda()
    names = fx.keys("sd")
    for s in range(1, 4):
        loss = fused_train_step(opt, x, t, "l2", fx.meta["clip"])
        assert abs(loss.item() - fx["losses"][s - 1]) < 2e-5 * max(1.0, fx["losses"][s - 1])
        got = np.concatenate([p.detach().cpu().numpy().reshape(-1) for p in net.parameters()])
        assert rel_err(got, fx.flat(f"p{s}", names)) < 3e-5, s


@pytest.mark.parametrize("bb,H,thx,thh", [("deltagru", 15, 0.0, 0.0), ("deltagru", 15, 0.01, 0.05), ("deltagru", 8, 0.02, 0.1),
                                          ("deltagru", 24, 0.01, 0.05), ("deltagru_tcnskip", 15, 0.01, 0.05),
                                          ("deltagru_tcnskip", 16, 0.0, 0.0), ("deltagru_tcnskip", 9, 0.05, 0.02),
                                          ("deltagru_tcnskip", 32, 0.01, 0.02)])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (17, 32), (7, 33), (5, 200), (66, 63), (2, 97)])
def test_dx_against_oracle(bb, H, thx, thh, B, T):
    """dL/dx of the delta backbones (frozen PA of a cascade / x.requires_grad): through W_ih^T, the keep / carry logic of
    x_p, the feature Jacobian — for TRes also the next-sample features (torch.roll wrap included) and the TCN skip path —
    against the oracle, with and without the weight gradients in the same backward."""
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H * 100 + B + T)
    net = CoreModel(2, H, 1, bb, thx=thx, thh=thh).cuda()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    rng = np.random.RandomState(B * 13 + T)
    amp = 0.05 + 0.85 * rng.rand(B, T, 1)
    ph = 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    dy = rng.randn(B, T, 2).astype(np.float32)
    net.backbone.set_debug(1)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    y = net(xt)
    y.backward(torch.from_numpy(dy).cuda())
    o = Oracle("f32")
    m = make_model(bb, H, thx, thh)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    yo, so = o.forward(m, p, x)
    go, dxo = o.backward(m, p, x, dy, need_dx=True)
    g = np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in net.parameters()])
    st = net.backbone.statistics
    assert abs(st["num_dx_zeros"] - so[0]) <= 2 and abs(st["num_dh_zeros"] - so[2]) <= 2
    exact = st["num_dx_zeros"] == so[0] and st["num_dh_zeros"] == so[2]
    tol_f, tol_g = (FWD_TOL, GRAD_TOL) if exact else (5e-3, 5e-2)
    assert rel_err(y.detach().cpu().numpy(), yo) < tol_f
    assert rel_err(g, go) < tol_g
    assert rel_err(xt.grad.cpu().numpy(), dxo) < tol_g
    # frozen model: dL/dx alone
    for q in net.parameters():
        q.requires_grad_(False)
    xt2 = torch.from_numpy(x).cuda().requires_grad_(True)
    net(xt2).backward(torch.from_numpy(dy).cuda())
    assert rel_err(xt2.grad.cpu().numpy(), dxo) < tol_g


@pytest.mark.parametrize("name,bb", CASES)
def test_dx_golden(name, bb):
    """dL/dx on the reference-generated fixtures (gx of oracle/gen_golden.py)"""
    fx = Fixture(name)
    net = _model(fx, bb)
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_(True)
    loss = torch.nn.functional.mse_loss(net(x), torch.from_numpy(fx["tgt"]).cuda())
    loss.backward()
    assert rel_err(x.grad.cpu().numpy(), fx["gx"]) < GRAD_TOL
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), fx["g/" + k]) < GRAD_TOL, k


# ---- S16 split kernels (csrc/delta_s16.hip), forced for every batch size with the tuning knob ------------------------
@pytest.fixture
def force_s16():
    from opendpd_amd import _lib
    lib = _lib.load()
    assert lib.odpd_set_tuning(b"s16_min_batch", 0) == 0
    yield
    lib.odpd_set_tuning(b"s16_min_batch", -1)


@pytest.mark.parametrize("name,bb", CASES)
def test_s16_golden_forward_backward_and_counters(force_s16, name, bb):
    """S16 kernels on the golden inputs: outputs, gradients and — on the dense (threshold 0) fixtures and on these
    thresholded inputs — the exact sparsity counters (the MFMA summation order differs from ATen's, so a decision
    within rounding of a threshold could in principle flip; on the golden inputs none does)."""
    test_golden_forward_backward_and_counters(name, bb)


@pytest.mark.parametrize("bb,H,thx,thh", [("deltagru", 15, 0.0, 0.0), ("deltagru", 15, 0.01, 0.05), ("deltagru", 8, 0.02, 0.1),
                                          ("deltagru_tcnskip", 15, 0.01, 0.05), ("deltagru_tcnskip", 16, 0.0, 0.0),
                                          ("deltagru_tcnskip", 9, 0.05, 0.02)])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (17, 32), (7, 33), (5, 200), (66, 63)])
def test_s16_against_oracle_ragged(force_s16, bb, H, thx, thh, B, T):
    test_against_oracle_ragged(bb, H, thx, thh, B, T)


@pytest.mark.parametrize("name,bb", [("deltagru_h15_th", "deltagru"), ("tres_h15_th", "deltagru_tcnskip")])
def test_s16_train_steps_follow_reference(force_s16, name, bb):
    test_train_steps_follow_reference(name, bb)


def test_s16_cascade_config3_follows_reference(force_s16):
    """BASELINE config 3 with every kernel in the S16 mapping: TRes-DeltaGRU (delta_s16) -> frozen DGRU H23 (gru_s16n)"""
    from tests import test_cascade_gpu as casc
    casc.test_cascade_autograd_matches_reference("cascade_tres15_dgru23", "deltagru_tcnskip", "dgru")
    casc.test_cascade_fused_steps_follow_reference("cascade_tres15_dgru23", "deltagru_tcnskip", "dgru")


# ---- hidden 17..32: two unit tiles of the S16 mapping, selected at every batch size ---------------------------------
@pytest.mark.parametrize("bb,H,thx,thh", [("deltagru", 17, 0.0, 0.0), ("deltagru", 24, 0.01, 0.05), ("deltagru", 32, 0.02, 0.1),
                                          ("deltagru_tcnskip", 23, 0.01, 0.05), ("deltagru_tcnskip", 32, 0.0, 0.0)])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (17, 32), (7, 33), (5, 200), (66, 63)])
def test_wide_hidden_against_oracle_ragged(bb, H, thx, thh, B, T):
    test_against_oracle_ragged(bb, H, thx, thh, B, T)


def test_hidden_above_64_is_refused_loudly():
    """the kernels themselves refuse a hidden size beyond their envelope — 64 units since r04 (csrc/delta_wide.hip) — (the registry routes
    such a configuration to the ATen restatement instead, with a warning: tests/test_wide_cpu.py)"""
    from opendpd_amd import backbones as B
    net = B.DeltaGRU(input_size=6, hidden_size=65, output_size=2, num_layers=1).cuda()
    with pytest.raises(RuntimeError):
        net(torch.rand(2, 16, 2, device="cuda"))
    net = B.DeltaJANET(input_size=6, hidden_size=65, output_size=2, num_layers=1).cuda()
    with pytest.raises(RuntimeError):
        net(torch.rand(2, 16, 2, device="cuda"))


def test_dx_flag_does_not_leak_between_autograd_and_fused_calls():
    """ODPD_FLAG_NEED_DX is a per-call kernel selection: an autograd call that asked for dL/dx must not change which kernels
    (and checkpoint layout) the fused trainer of the same module uses afterwards — the steps stay those of the reference."""
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    fx = Fixture("tres_h15_th")
    net = _model(fx, "deltagru_tcnskip")
    opt = FusedAdamW(net, lr=fx.meta["lr"])
    x = torch.from_numpy(fx["x"]).cuda()
    t = torch.from_numpy(fx["tgt"]).cuda()
    names = fx.keys("sd")
    for s in range(1, 4):
        xg = x.clone().requires_grad_(True)            # an unrelated autograd pass with dL/dx in between
        net(xg).sum().backward()
        for p in net.parameters():
            p.grad = None
        assert net.backbone.desc.flags & 2 == 0
        loss = fused_train_step(opt, x, t, "l2", fx.meta["clip"])
        assert abs(loss.item() - fx["losses"][s - 1]) < 2e-5 * max(1.0, fx["losses"][s - 1])
        got = np.concatenate([p.detach().cpu().numpy().reshape(-1) for p in net.parameters()])
        assert rel_err(got, fx.flat(f"p{s}", names)) < 3e-5, s


@pytest.mark.parametrize("bb,H,thx,thh", [("deltagru", 15, 0.0, 0.0), ("deltagru", 15, 0.01, 0.05), ("deltagru", 8, 0.02, 0.1), ("deltagru", 1, 0.01, 0.05),
                                          ("deltagru_tcnskip", 15, 0.01, 0.05), ("deltagru_tcnskip", 16, 0.0, 0.0),
                                          ("deltagru_tcnskip", 9, 0.05, 0.02)])
@pytest.mark.parametrize("B,T", [(1, 700), (3, 2560), (2, 256), (8, 257)])
def test_evaluation_kernel_matches_the_oracle_and_keeps_the_counters(bb, H, thx, thh, B, T):
    """inference on a few long sequences (net_eval / run_dpd shapes; torch.no_grad(), so no checkpoints are asked for) runs
    delta_eval_kernel (one sequence per wave; rows r / z / n, features and the TRes skip computed per chunk with lane = time step, the
    x-side delta memory one feature per lane): the same thresholded arithmetic as the row-rotated forward and the oracle — sparsity counters
    and outputs (a rounding-level difference can flip a threshold decision; a handful of flips are allowed for)"""
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H * 10 + B)
    net = CoreModel(2, H, 1, bb, thx=thx, thh=thh).cuda().eval()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    g = torch.Generator().manual_seed(T)
    amp, ph = 0.05 + 0.85 * torch.rand(B, T, 1, generator=g), 2 * np.pi * torch.rand(B, T, 1, generator=g)
    x = torch.cat((amp * torch.cos(ph), amp * torch.sin(ph)), -1)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    yo, so = Oracle("f32").forward(make_model(bb, H, thx, thh), p, x.numpy())
    keys = ("num_dx_zeros", "num_dx_numel", "num_dh_zeros", "num_dh_numel")
    net.backbone.set_debug(1)
    with torch.no_grad():
        y_eval = net(x.cuda()).cpu().numpy()
    st_eval = [net.backbone.statistics[k] for k in keys]
    import ctypes as C
    from opendpd_amd import _lib
    _lib.load().odpd_set_tuning(b"gp_max_batch", C.c_int64(0))                 # one-sequence-per-wave kernels off: the row-rotated forward
    try:
        net.backbone.set_debug(1)
        y_train = net(x.cuda().requires_grad_(True)).detach().cpu().numpy()
        st_train = [net.backbone.statistics[k] for k in keys]
    finally:
        _lib.load().odpd_set_tuning(b"gp_max_batch", C.c_int64(-1))
    assert st_eval[1] == st_train[1] and st_eval[3] == st_train[3]
    same = st_eval == st_train              # the recurrent sums run in a different order: a |dh| within rounding of th_h can flip a mask
    assert abs(st_eval[0] - st_train[0]) + abs(st_eval[2] - st_train[2]) <= 4
    assert rel_err(y_eval, y_train) < (FWD_TOL if same else 5e-3)      # the accumulators integrate rounding over T steps
    assert st_eval[1] == so[1] and st_eval[3] == so[3]
    flips = abs(st_eval[0] - so[0]) + abs(st_eval[2] - so[2])
    assert flips <= 4
    assert rel_err(y_eval, yo) < (FWD_TOL if flips == 0 else 5e-3)


@pytest.mark.parametrize("bb,H,thx,thh", [("deltagru", 15, 0.01, 0.05), ("deltagru", 8, 0.0, 0.0), ("deltagru", 1, 0.02, 0.1),
                                          ("deltagru_tcnskip", 15, 0.01, 0.05), ("deltagru_tcnskip", 16, 0.0, 0.0), ("deltagru_tcnskip", 9, 0.05, 0.02)])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 17), (7, 63), (5, 64), (2, 65), (64, 50), (9, 200), (600, 200)])
def test_gate_parallel_backward_kernel(bb, H, thx, thh, B, T):
    """the split train path at the reference's batch sizes runs delta_gp_bwd_kernel (one sequence per wave; it runs the forward again and
    parks the step state in LDS instead of reading checkpoints; (600, 200): several sequences per workgroup): parameter gradients against
    the oracle and against the four-sequence-per-wave backward (odpd_set_tuning gp_max_batch = 0) of the same batch"""
    import ctypes as C
    from opendpd_amd import CoreModel, _lib
    from oracle.oracle import Oracle, make_model
    lib = _lib.load()
    torch.manual_seed(H * 100 + B + T)
    net = CoreModel(2, H, 1, bb, thx=thx, thh=thh).cuda()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    rng = np.random.RandomState(B * 13 + T)
    amp, ph = 0.05 + 0.85 * rng.rand(B, T, 1), 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    dy = rng.randn(B, T, 2).astype(np.float32)
    got = {}
    try:
        for gp in (1 << 30, 0):      # one-sequence-per-wave kernels forced on / off
            lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(gp))
            for q in net.parameters():
                q.grad = None
            net(torch.from_numpy(x).cuda()).backward(torch.from_numpy(dy).cuda())
            got[gp] = torch.cat([q.grad.reshape(-1) for q in net.parameters()]).cpu().numpy()
    finally:
        lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(-1))
    # thresholded: the two mappings sum in different orders, a |dh| within rounding of th_h can flip a mask somewhere in B x T x H decisions
    assert rel_err(got[1 << 30], got[0]) < (2e-5 if thx == 0.0 and thh == 0.0 else 5e-3)
    assert T < 50 or not np.array_equal(got[1 << 30], got[0])          # two kernels
    if B * T <= 3000:
        o, m = Oracle("f32"), make_model(bb, H, thx, thh)
        p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
        _, so = o.forward(m, p, x)
        go, _ = o.backward(m, p, x, dy, need_dx=False)
        # a rounding-level difference can flip a threshold decision (see test_against_oracle_ragged)
        assert rel_err(got[1 << 30], go) < (GRAD_TOL if thx == 0.0 and thh == 0.0 else 5e-2)


@pytest.mark.parametrize("bb,H,thx,thh", [("deltagru", 15, 0.01, 0.05), ("deltagru", 8, 0.0, 0.0), ("deltagru_tcnskip", 15, 0.01, 0.05), ("deltagru_tcnskip", 9, 0.05, 0.02),
                                          ("deltajanet", 15, 0.0, 0.0), ("deltajanet", 7, 0.0, 0.0)])
@pytest.mark.parametrize("B,T,loss", [(1, 1, "l2"), (3, 17, "l2"), (7, 65, "l1"), (64, 50, "l2"), (9, 200, "l2"), (256, 200, "l2"), (600, 50, "l1")])
def test_fused_train_step_equals_the_split_chain(bb, H, thx, thh, B, T, loss):
    """r04: the train_pa step of a delta backbone at the reference's batch sizes as ONE launch (delta_gp_bwd_kernel<.., FUSED>: its forward
    pass counts the statistics, y / loss / dL/dy are formed inside with lane = time step) against the chain it replaces — evaluation
    kernel, loss kernel, the same backward kernel fed with dL/dy: bit-identical gradients and parameters (same arithmetic in the same
    order; TRes to 1e-7: its Hardswish is contracted differently), equal counters; loss and gradients against the oracle."""
    from opendpd_amd import CoreModel, _lib
    from opendpd_amd.train_funcs import FusedAdamW, _cascade_train_step, fused_train_step
    from oracle.oracle import Oracle, make_model
    kw = dict(thx=thx, thh=thh) if bb != "deltajanet" else {}
    rng = np.random.RandomState(B * 13 + T)
    amp, ph = 0.05 + 0.85 * rng.rand(B, T, 1), 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    t = (0.5 * rng.randn(B, T, 2)).astype(np.float32)
    out = []
    for fused in (True, False):
        torch.manual_seed(H * 100 + B + T)
        net = CoreModel(2, H, 1, bb, **kw).cuda()
        with torch.no_grad():
            for k, p in net.named_parameters():
                if "bias" in k:
                    p.uniform_(-0.3, 0.3)
        p0 = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
        net.backbone.set_debug(1)
        opt = FusedAdamW(net, lr=1e-3)
        assert opt.has_fused(B, T)
        xt, tt = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
        lv = fused_train_step(opt, xt, tt, loss, 200.0) if fused else _cascade_train_step(opt, xt, tt, loss, 200.0, B * T * 2)
        torch.cuda.synchronize()
        st = net.backbone.statistics
        out.append((float(lv.item()), opt.grad.cpu().numpy().copy(), net.backbone.flat_params().cpu().numpy().copy(),
                    [st["num_dx_zeros"], st["num_dx_numel"], st["num_dh_zeros"], st["num_dh_numel"]]))
    P = len(p0)
    if B <= 512 and bb != "deltagru_tcnskip":      # (beyond 2 x CUs frames the chain's forward is the four-sequences-per-wave kernel: same values, another summation order)
        assert np.array_equal(out[0][1][:P], out[1][1][:P]) and np.array_equal(out[0][2], out[1][2])
    elif B <= 512:      # TRes: the compiler contracts Hardswish differently where the skip, its gradient and the loss meet in one kernel (1e-7 relative)
        assert rel_err(out[0][1][:P], out[1][1][:P]) < (2e-6 if thx == 0.0 and thh == 0.0 else 5e-3) and rel_err(out[0][2], out[1][2]) < 1e-5
    else:
        assert rel_err(out[0][1][:P], out[1][1][:P]) < (2e-5 if thx == 0.0 and thh == 0.0 else 5e-3)
    assert out[0][3][1] == out[1][3][1] == 6 * B * T and out[0][3][3] == out[1][3][3] == H * B * T
    if B <= 512:
        assert out[0][3] == out[1][3]
    else:
        assert all(abs(a - b) <= 1e-4 * max(b, 1) for a, b in zip(out[0][3], out[1][3]))
    assert abs(out[0][0] - out[1][0]) <= 1e-6 * abs(out[1][0])
    if B * T <= 3000:
        o, m = Oracle("f32"), make_model(bb, H, *((thx, thh) if bb != "deltajanet" else ()))
        yo, _ = o.forward(m, p0, x)
        lo, dy = o.loss(loss, yo, t)
        go, _ = o.backward(m, p0, x, dy, need_dx=False)
        dense = thx == 0.0 and thh == 0.0
        assert abs(out[0][0] - lo) <= (2e-5 if dense else 2e-2) * max(1.0, abs(lo))
        assert rel_err(out[0][1][:P], go) < (GRAD_TOL if dense else 5e-2)


def test_fused_delta_step_reads_frames_in_place_and_runs_the_native_epoch_loop():
    """delta backbones now have a frame-reading fused kernel at the reference's batch sizes: net_train drives whole epochs through
    odpd_train_epoch (the `workspace` argument carrying the sparsity counters) — same result as the per-step Python loop over gathered frames"""
    import tests.test_e2e_gpu as e2e
    e2e.test_native_epoch_loop_equals_per_step_loop("deltagru_tcnskip", 15, 50, 64)
    e2e.test_native_epoch_loop_equals_per_step_loop("deltagru", 8, 200, 256)
