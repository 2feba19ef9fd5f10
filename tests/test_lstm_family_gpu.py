"""GPU parity of the LSTM-family kernels (lstm, vdlstm) against the reference golden vectors and
the CPU oracle (ragged sizes), plus the split-kernel train step trajectory."""
import numpy as np
import pytest
import torch

from tests.golden_util import Fixture, rel_err

pytestmark = pytest.mark.gpu
FWD_TOL, GRAD_TOL = 2e-5, 3e-4


def _model(fx, bb):
    from opendpd_amd import CoreModel
    net = CoreModel(2, fx.meta["hidden"], 1, bb)
    net.load_state_dict({k: torch.from_numpy(fx["sd/" + k]) for k in fx.keys("sd")})
    return net.cuda()


@pytest.mark.parametrize("name,bb", [("lstm_h14", "lstm"), ("vdlstm_h13", "vdlstm")])
def test_golden_forward_backward(name, bb):
    fx = Fixture(name)
    net = _model(fx, bb)
    need_dx = True
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_(need_dx)
    y = net(x)
    assert rel_err(y.detach().cpu().numpy(), fx["y"]) < FWD_TOL
    loss = torch.nn.functional.mse_loss(y, torch.from_numpy(fx["tgt"]).cuda())
    assert abs(loss.item() - fx["losses"][0]) < 1e-5 * max(1.0, fx["losses"][0])
    loss.backward()
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), fx["g/" + k]) < GRAD_TOL, k
    if need_dx:
        assert rel_err(x.grad.cpu().numpy(), fx["gx"]) < GRAD_TOL
    with torch.no_grad():
        ya = net(torch.from_numpy(fx["xa"]).cuda())
    assert rel_err(ya.cpu().numpy(), fx["ya"]) < FWD_TOL


@pytest.mark.parametrize("bb,H", [("lstm", 14), ("lstm", 9), ("lstm", 20), ("vdlstm", 13), ("vdlstm", 8), ("vdlstm", 18)])
@pytest.mark.parametrize("B,T", [(1, 3), (3, 5), (4, 32), (7, 33), (5, 200), (66, 63)])
def test_against_oracle_ragged(bb, H, B, T):
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H * 100 + B + T)
    net = CoreModel(2, H, 1, bb).cuda()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    rng = np.random.RandomState(B * 11 + T)
    amp = 0.05 + 0.85 * rng.rand(B, T, 1)
    ph = 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    dy = rng.randn(B, T, 2).astype(np.float32)
    need_dx = True
    xt = torch.from_numpy(x).cuda().requires_grad_(need_dx)
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
    if need_dx:
        assert rel_err(xt.grad.cpu().numpy(), dxo) < GRAD_TOL


@pytest.mark.parametrize("s16", [False, True])
@pytest.mark.parametrize("H,B,T", [(14, 3, 2), (23, 16, 1), (27, 3, 2), (9, 1, 1)])
def test_lstm_frames_shorter_than_the_halo(s16, H, B, T):
    """plain lstm shares the 3-sample halo staging with vdlstm but never uses it: frames with T < 3 are legal and must
    neither read before the frame start (found by tools/oob_hunt.py) nor differ from the oracle."""
    from opendpd_amd import CoreModel, _lib
    from oracle.oracle import Oracle, make_model
    lib = _lib.load()
    lib.odpd_set_tuning(b"s16_min_batch", 0 if s16 else -1)
    try:
        torch.manual_seed(H + B + T)
        net = CoreModel(2, H, 1, "lstm").cuda()
        rng = np.random.RandomState(H)
        x = (rng.uniform(0.05, 0.9, (B, T, 2)) * rng.choice([-1.0, 1.0], (B, T, 2))).astype(np.float32)
        dy = rng.randn(B, T, 2).astype(np.float32)
        xt = torch.from_numpy(x).cuda().requires_grad_(True)
        y = net(xt)
        y.backward(torch.from_numpy(dy).cuda())
    finally:
        lib.odpd_set_tuning(b"s16_min_batch", -1)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    m = make_model("lstm", H)
    yo, _ = Oracle("f32").forward(m, p, x)
    go, dxo = Oracle("f32").backward(m, p, x, dy)
    g = np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in net.parameters()])
    assert rel_err(y.detach().cpu().numpy(), yo) < FWD_TOL
    assert rel_err(g, go) < GRAD_TOL
    assert rel_err(xt.grad.cpu().numpy(), dxo) < GRAD_TOL


@pytest.mark.parametrize("name,bb", [("lstm_h14", "lstm"), ("vdlstm_h13", "vdlstm")])
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


@pytest.mark.parametrize("H,B,T", [(13, 5, 37), (8, 2, 3), (20, 9, 200)])
def test_vdlstm_frozen_pa_gives_dx_only(H, B, T):
    """vdlstm as the frozen PA of a cascade (models.py:169-171): dL/dx alone, circular-halo samples included."""
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H + T)
    net = CoreModel(2, H, 1, "vdlstm").cuda()
    for p in net.parameters():
        p.requires_grad_(False)
    rng = np.random.RandomState(T)
    x = (rng.uniform(0.05, 0.9, (B, T, 2)) * rng.choice([-1.0, 1.0], (B, T, 2))).astype(np.float32)
    dy = rng.randn(B, T, 2).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    net(xt).backward(torch.from_numpy(dy).cuda())
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    _, dxo = Oracle("f32").backward(make_model("vdlstm", H), p, x, dy)
    assert rel_err(xt.grad.cpu().numpy(), dxo) < GRAD_TOL
    assert all(q.grad is None for q in net.parameters())


def test_vdlstm_short_frame_is_refused_loudly():
    """the reference pads with the frame's own last 3 samples (vdlstm.py:66-74): T < 3 has no meaning"""
    from opendpd_amd import CoreModel
    net = CoreModel(2, 8, 1, "vdlstm").cuda()
    with pytest.raises(RuntimeError):
        net(torch.rand(2, 2, 2, device="cuda"))


# ---- S16 fused train kernel (csrc/lstm_s16.hip), forced for every batch size with the tuning knob ----------------
@pytest.fixture
def force_s16():
    from opendpd_amd import _lib
    lib = _lib.load()
    assert lib.odpd_set_tuning(b"s16_min_batch", 0) == 0
    yield
    lib.odpd_set_tuning(b"s16_min_batch", -1)


@pytest.mark.parametrize("name,bb", [("lstm_h14", "lstm"), ("vdlstm_h13", "vdlstm")])
def test_s16_fused_steps_follow_reference(force_s16, name, bb):
    from opendpd_amd.train_funcs import FusedAdamW
    fx = Fixture(name)
    opt = FusedAdamW(_model(fx, bb), lr=fx.meta["lr"])
    assert opt.has_fused(fx["x"].shape[0], fx["x"].shape[1])          # the single-launch kernel is the one that runs
    test_train_steps_follow_reference(name, bb)


@pytest.mark.parametrize("bb,H,B,T", [("lstm", 14, 256, 200), ("vdlstm", 13, 37, 50), ("lstm", 9, 1027, 64), ("vdlstm", 8, 3, 333),
                                      ("vdlstm", 16, 17, 3), ("lstm", 23, 40, 70), ("vdlstm", 20, 33, 65), ("lstm", 1, 16, 4)])
def test_s16_fused_equals_unfused_gradients(force_s16, bb, H, B, T):
    """single-launch S16 step (L2 and L1) == autograd through the row-rotated split kernels (oracle-checked above)"""
    from opendpd_amd import CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    torch.manual_seed(1)
    net = CoreModel(2, H, 1, bb).cuda()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.6
    x = x + 0.05 * torch.sign(x)
    t = torch.randn(B, T, 2, device="cuda", generator=g) * 0.3
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    assert opt.has_fused(B, T)
    for kind, fn in (("l2", torch.nn.functional.mse_loss), ("l1", torch.nn.functional.l1_loss)):
        for p in net.parameters():
            p.grad = None
        loss = fn(net(x), t)
        loss.backward()
        gref = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
        lf = fused_train_step(opt, x, t, kind, 0.0)
        assert abs(lf.item() - loss.item()) < 1e-5 * max(1.0, loss.item()), kind
        assert rel_err(opt.grad[:-4].cpu().numpy(), gref) < 3e-5, kind


@pytest.mark.parametrize("bb", ["lstm", "vdlstm"])
@pytest.mark.parametrize("H", [1, 8, 13, 16, 17, 23, 32])
@pytest.mark.parametrize("B,T", [(1, 700), (3, 2560), (2, 256), (8, 257)])
def test_evaluation_kernel_matches_the_oracle(bb, H, B, T):
    """inference on a few long sequences (net_eval / run_dpd shapes; torch.no_grad(), so no checkpoints are asked for) runs the
    gate-parallel evaluation kernel (lstm_eval_kernel: one sequence per wave, row k of the wave computes gate k, three cross-row swaps
    hand every row all four; hidden 17..32 as two unit blocks per row): against the oracle, and against the row-rotated forward the same
    call takes when gradients are enabled"""
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H * 10 + B)
    net = CoreModel(2, H, 1, bb).cuda().eval()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    g = torch.Generator().manual_seed(T)
    amp, ph = 0.05 + 0.85 * torch.rand(B, T, 1, generator=g), 2 * np.pi * torch.rand(B, T, 1, generator=g)
    x = torch.cat((amp * torch.cos(ph), amp * torch.sin(ph)), -1)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    yo, _ = Oracle("f32").forward(make_model(bb, H), p, x.numpy())
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


@pytest.mark.parametrize("bb", ["lstm", "vdlstm"])
@pytest.mark.parametrize("H", [1, 5, 13, 16])
@pytest.mark.parametrize("B,T", [(1, 3), (3, 5), (7, 63), (5, 64), (2, 65), (64, 50), (9, 200), (300, 200), (5, 130)])
def test_gate_parallel_train_kernel(bb, H, B, T):
    """the reference's own batch sizes run lstm_gp_train_kernel (one sequence per wave, row k = gate k; outputs, loss and the head's
    gradients with lane = time step; weight gradients as 4-block MFMAs; (300, 200) recomputes the gates, the smaller batches park them): loss
    and gradient against the oracle (L2 and L1), and against the split forward / loss / backward kernels (odpd_set_tuning gp_max_batch = 0)"""
    import ctypes as C
    from opendpd_amd import CoreModel, _lib
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    lib = _lib.load()
    torch.manual_seed(H * 100 + B + T)
    net = CoreModel(2, H, 1, bb).cuda()
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
    o, m = Oracle("f32"), make_model(bb, H)
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
            # the split kernels on the same batch
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


@pytest.mark.parametrize("bb,H,B,T", [("vdlstm", 13, 37, 50), ("lstm", 9, 33, 20), ("vdlstm", 5, 16, 7), ("lstm", 13, 17, 64), ("vdlstm", 1, 3, 5)])
def test_s16_k_packed_kernel_equals_the_unpacked_one(force_s16, bb, H, B, T):
    """hidden <= 13: lstm16_train_kernel<.., K-packed> (three input slots in the h tile's free K positions, their gradients out of the recurrent
    gradient tiles; vdlstm's amplitude-0 / bias gradients on the VALU) against the unpacked instantiation (odpd_set_tuning("lstm_pack", 0)):
    the same products in another summation order"""
    from opendpd_amd import CoreModel, _lib
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    lib = _lib.load()
    torch.manual_seed(3)
    net = CoreModel(2, H, 1, bb).cuda()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    g = torch.Generator(device="cuda").manual_seed(7)
    x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.6
    x = x + 0.05 * torch.sign(x)
    t = torch.randn(B, T, 2, device="cuda", generator=g) * 0.3
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    assert opt.has_fused(B, T)
    out = {}
    try:
        for pack in (1, 0):
            assert lib.odpd_set_tuning(b"lstm_pack", pack) == 0
            loss = fused_train_step(opt, x, t, "l2", 0.0)
            out[pack] = (loss.item(), opt.grad[:-4].cpu().numpy().copy())
    finally:
        lib.odpd_set_tuning(b"lstm_pack", 1)
    assert abs(out[1][0] - out[0][0]) < 1e-6 * max(1.0, out[0][0])
    assert rel_err(out[1][1], out[0][1]) < 5e-6
    assert np.abs(out[0][1]).max() > 0
