"""GPU parity of the PGJANET, TCNN and NeuralTX (the NTX instantiation of csrc/tcnn.hip) kernels against the reference golden
vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from tests.golden_util import Fixture, rel_err

pytestmark = pytest.mark.gpu
FWD_TOL, GRAD_TOL = 2e-5, 3e-4
GOLDEN = [("pgjanet_h11", "pgjanet"), ("tcnn_c35", "tcnn"), ("neuraltx_c36", "neuraltx"), ("neuraltx_c12", "neuraltx")]


def _supported(bb):
    from opendpd_amd.models import CoreModel
    try:
        CoreModel(2, 8, 1, bb)
        return True
    except NotImplementedError:
        return False


def _model(fx, bb):
    from opendpd_amd import CoreModel
    net = CoreModel(2, fx.meta["hidden"], 1, bb)
    net.load_state_dict({k: torch.from_numpy(fx["sd/" + k]) for k in fx.keys("sd")})
    return net.cuda()


@pytest.mark.parametrize("name,bb", GOLDEN)
def test_golden_forward_backward(name, bb):
    if not _supported(bb):
        pytest.skip(f"{bb} kernel not built yet")
    fx = Fixture(name)
    net = _model(fx, bb)
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
        ya = net(torch.from_numpy(fx["xa"]).cuda())
    assert rel_err(ya.cpu().numpy(), fx["ya"]) < FWD_TOL


@pytest.mark.parametrize("bb,H", [("pgjanet", 11), ("pgjanet", 8), ("pgjanet", 16), ("tcnn", 35), ("tcnn", 8), ("tcnn", 30),
                                   ("neuraltx", 36), ("neuraltx", 8), ("neuraltx", 64)])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (4, 32), (7, 33), (5, 200), (66, 63), (2, 2100), (3, 257), (1, 700)])
def test_against_oracle_ragged(bb, H, B, T):
    if not _supported(bb):
        pytest.skip(f"{bb} kernel not built yet")
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H * 100 + B + T)
    net = CoreModel(2, H, 1, bb).cuda()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
            if k.startswith("backbone.conv_"):        # NeuralTX: FIR taps of a size that makes every path count (init gain is 0.1)
                p.uniform_(-0.6, 0.6)
    rng = np.random.RandomState(B * 17 + T)
    amp = 0.05 + 0.85 * rng.rand(B, T, 1)
    ph = 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    dy = rng.randn(B, T, 2).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)     # (2, 2100): TCNN time tiles with halos, several BPTT chunks
    y = net(xt)
    y.backward(torch.from_numpy(dy).cuda())
    o = Oracle("f32")
    m = make_model(bb, H)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    yo, _ = o.forward(m, p, x)
    go, dxo = o.backward(m, p, x, dy)
    g = np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in net.parameters()])
    assert rel_err(y.detach().cpu().numpy(), yo) < FWD_TOL
    assert rel_err(g, go) < GRAD_TOL
    assert rel_err(xt.grad.cpu().numpy(), dxo) < GRAD_TOL
    # frozen model (PA of a cascade): dL/dx alone
    for q in net.parameters():
        q.requires_grad_(False)
        q.grad = None
    xt2 = torch.from_numpy(x).cuda().requires_grad_(True)
    net(xt2).backward(torch.from_numpy(dy).cuda())
    assert rel_err(xt2.grad.cpu().numpy(), dxo) < GRAD_TOL


@pytest.mark.parametrize("name,bb", GOLDEN)
def test_train_steps_follow_reference(name, bb):
    if not _supported(bb):
        pytest.skip(f"{bb} kernel not built yet")
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


# ---- PGJANET S16 kernels (csrc/janet_s16.hip), forced for every batch size with the tuning knob -----------------------
@pytest.fixture
def force_s16():
    from opendpd_amd import _lib
    lib = _lib.load()
    assert lib.odpd_set_tuning(b"s16_min_batch", 0) == 0
    yield
    lib.odpd_set_tuning(b"s16_min_batch", -1)


def test_s16_pgjanet_golden(force_s16):
    test_golden_forward_backward("pgjanet_h11", "pgjanet")
    test_train_steps_follow_reference("pgjanet_h11", "pgjanet")


@pytest.mark.parametrize("H", [11, 8, 16])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (17, 32), (7, 33), (5, 200), (66, 63)])
def test_s16_pgjanet_against_oracle_ragged(force_s16, H, B, T):
    test_against_oracle_ragged("pgjanet", H, B, T)


def test_s16_pgjanet_keeps_relative_accuracy_at_small_arguments(force_s16):
    """PGJANET's tanh arguments are small (zero biases at init, u <= 1/64): against the float64 oracle the S16 kernel must
    stay at rounding level — the single-formula tanh of the other S16 kernels lost 2e-5 here (tools/mapping_crosscheck.py)."""
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    B, T, H = 64, 200, 11
    g = torch.Generator().manual_seed(0)
    amp, ph = 0.05 + 0.85 * torch.rand(B, T, 1, generator=g), 2 * np.pi * torch.rand(B, T, 1, generator=g)
    x = torch.cat((amp * torch.cos(ph), amp * torch.sin(ph)), -1)
    torch.manual_seed(0)
    net = CoreModel(2, H, 1, "pgjanet").cuda()
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    yo, _ = Oracle("f64").forward(make_model("pgjanet", H), p.astype(np.float64), x.numpy().astype(np.float64))
    with torch.no_grad():
        y = net(x.cuda()).cpu().numpy()
    assert rel_err(y, yo) < 2e-6


@pytest.mark.parametrize("H", [1, 8, 11, 13, 16])
@pytest.mark.parametrize("B,T", [(1, 700), (3, 2560), (2, 256), (8, 257)])
def test_pgjanet_evaluation_kernel_matches_the_oracle(H, B, T):
    """inference on a few long sequences (net_eval / run_dpd shapes; torch.no_grad(), so no checkpoints are asked for) runs
    janet_eval_kernel (one sequence per wave, the step's seven H x H products as two rounds of one rotated dot product per row):
    against the oracle, and against the row-rotated forward the same call takes when gradients are enabled"""
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H * 10 + B)
    net = CoreModel(2, H, 1, "pgjanet").cuda().eval()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    g = torch.Generator().manual_seed(T)
    amp, ph = 0.05 + 0.85 * torch.rand(B, T, 1, generator=g), 2 * np.pi * torch.rand(B, T, 1, generator=g)
    x = torch.cat((amp * torch.cos(ph), amp * torch.sin(ph)), -1)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    yo, _ = Oracle("f32").forward(make_model("pgjanet", H), p, x.numpy())
    import ctypes as C
    from opendpd_amd import _lib
    with torch.no_grad():
        y_eval = net(x.cuda()).cpu().numpy()
    y_ckpt = net(x.cuda().requires_grad_(True)).detach().cpu().numpy()         # gradients enabled: the same kernel also writes the BPTT checkpoints
    _lib.load().odpd_set_tuning(b"gp_max_batch", C.c_int64(0))                 # one-sequence-per-wave kernels off: the row-rotated forward
    try:
        y_train = net(x.cuda().requires_grad_(True)).detach().cpu().numpy()
    finally:
        _lib.load().odpd_set_tuning(b"gp_max_batch", C.c_int64(-1))
    assert rel_err(y_eval, yo) < FWD_TOL and rel_err(y_train, yo) < FWD_TOL
    assert np.array_equal(y_eval, y_ckpt)
    assert rel_err(y_eval, y_train) < 5e-6
    assert H == 1 or not np.array_equal(y_eval, y_train)          # two kernels: g's pre-activation is summed in a different order


@pytest.mark.parametrize("H", [1, 5, 11, 16])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (7, 63), (5, 64), (2, 65), (64, 50), (9, 200), (300, 200), (5, 130)])
def test_pgjanet_gate_parallel_train_kernel(H, B, T):
    """the reference's own batch sizes run janet_gp_train_kernel (one sequence per wave; two forward and two transposed rounds of one rotated
    dot product per row, the step's seven weight gradients as two 4-block MFMAs; fc_out, loss and dL/dy with lane = time step): loss and
    gradient against the oracle (L2 and L1), and against the split forward / loss / backward kernels (odpd_set_tuning gp_max_batch = 0)"""
    import ctypes as C
    from opendpd_amd import CoreModel, _lib
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    lib = _lib.load()
    torch.manual_seed(H * 100 + B + T)
    net = CoreModel(2, H, 1, "pgjanet").cuda()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    rng = np.random.RandomState(B * 13 + T)
    amp, ph = 0.05 + 0.85 * rng.rand(B, T, 1), 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    tgt = (0.4 * rng.randn(B, T, 2)).astype(np.float32)
    xt, tt = torch.from_numpy(x).cuda(), torch.from_numpy(tgt).cuda()
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    o, m = Oracle("f32"), make_model("pgjanet", H)
    yo, _ = o.forward(m, p, x)
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    assert opt.has_fused(B, T)
    try:
        for kind in ("l2", "l1"):
            d = yo - tgt
            lo = float((d * d).mean()) if kind == "l2" else float(np.abs(d).mean())
            dy = (2 * d / d.size if kind == "l2" else np.sign(d) / d.size).astype(np.float32)
            go, _ = o.backward(m, p, x, dy, need_dx=False)
            loss = fused_train_step(opt, xt, tt, kind, 0.0)
            got = opt.grad[:-4].cpu().numpy().copy()
            assert abs(float(loss) - lo) < 2e-5 * max(1.0, lo)
            assert rel_err(got, go) < GRAD_TOL
            lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(0))
            opt2 = FusedAdamW(net, lr=0.0, weight_decay=0.0)
            assert not opt2.has_fused(B, T)
            for q in net.parameters():
                q.grad = None
            y = net(xt)
            l2 = torch.nn.functional.mse_loss(y, tt) if kind == "l2" else torch.nn.functional.l1_loss(y, tt)
            l2.backward()
            gs = torch.cat([q.grad.reshape(-1) for q in net.parameters()]).cpu().numpy()
            assert abs(float(loss) - l2.item()) < 1e-5 * max(1.0, lo) and rel_err(got, gs) < 2e-5
            lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(-1))
    finally:
        lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(-1))
