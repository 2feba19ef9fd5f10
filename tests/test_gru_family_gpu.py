"""GPU parity tests of the GRU-family HIP kernels (gru, dgru, qgru, qgru_amp1) through the registry
and through the raw C ABI: against the reference golden vectors (tests/golden) and against the CPU
oracle on seeded inputs, including ragged batch / time sizes.

Tolerances (fp32, stated relative to the max magnitude of the compared tensor):
  forward outputs 2e-5, gradients 2e-4 (BPTT over <= 333 steps; sums in a different order than ATen).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.golden_util import Fixture, rel_err

pytestmark = pytest.mark.gpu

FWD_TOL, GRAD_TOL = 2e-5, 2e-4
SINGLE = [("gru_h11", "gru"), ("gru_h23", "gru"), ("dgru_h13", "dgru"), ("dgru_h8", "dgru"), ("dgru_h23", "dgru"),
          ("qgru_h10", "qgru"), ("qgru_h16", "qgru"), ("qgru_amp1_h10", "qgru_amp1")]


def _model(fx, bb):
    from opendpd_amd import CoreModel
    net = CoreModel(2, fx.meta["hidden"], 1, bb)
    net.load_state_dict({k: torch.from_numpy(fx["sd/" + k]) for k in fx.keys("sd")})
    return net.cuda()


@pytest.mark.parametrize("name,bb", SINGLE)
def test_golden_forward_backward(name, bb):
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
    # config-shaped frames (T = 200)
    with torch.no_grad():
        ya = net(torch.from_numpy(fx["xa"]).cuda())
    assert rel_err(ya.cpu().numpy(), fx["ya"]) < FWD_TOL


@pytest.mark.parametrize("bb,H", [("gru", 11), ("dgru", 13), ("dgru", 23), ("gru", 16), ("dgru", 32), ("qgru", 10),
                                  ("qgru_amp1", 17)])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (4, 64), (7, 65), (5, 200), (130, 63), (2, 333)])
def test_against_oracle_ragged(bb, H, B, T):
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H * 1000 + B * 10 + T)
    net = CoreModel(2, H, 1, bb).cuda()
    with torch.no_grad():  # biases are zero after init: make them count
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    rng = np.random.RandomState(B * 7 + T)
    amp = 0.05 + 0.85 * rng.rand(B, T, 1)
    ph = 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    dy = rng.randn(B, T, 2).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
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


def test_frozen_model_gives_dx_only():
    """PA of a cascade: requires_grad=False on every parameter, gradient w.r.t. the input only."""
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(3)
    net = CoreModel(2, 23, 1, "dgru").cuda()
    for p in net.parameters():
        p.requires_grad = False
    rng = np.random.RandomState(1)
    x = (0.1 + 0.5 * rng.rand(6, 50, 2)).astype(np.float32)
    dy = rng.randn(6, 50, 2).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    net(xt).backward(torch.from_numpy(dy).cuda())
    assert all(p.grad is None for p in net.parameters())
    o = Oracle("f32")
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    _, dxo = o.backward(make_model("dgru", 23), p, x, dy)
    assert rel_err(xt.grad.cpu().numpy(), dxo) < GRAD_TOL


def test_issue_probe_reports_a_plausible_speed_and_rejects_bad_arguments():
    """odpd_probe_issue_ns (bench.py's `config.issue_probe_ns`): ns per wave instruction per SIMD of a pure v_fmac loop at four waves per SIMD —
    1.74 on the MI355X boxes of r06 (four cycles of a 16-lane SIMD at 2.3 GHz); never below half of that floor at 3 GHz, never tens of ns."""
    from opendpd_amd import _lib
    lib = _lib.load()
    ns = C.c_double(0.0)
    assert lib.odpd_probe_issue_ns(_lib.stream_ptr(), 4000, C.byref(ns)) == 0
    assert 0.4 < ns.value < 6.0, ns.value
    assert lib.odpd_probe_issue_ns(_lib.stream_ptr(), 0, C.byref(ns)) == -1
    assert lib.odpd_probe_issue_ns(_lib.stream_ptr(), 10, None) == -1


def test_c_abi_direct_and_errors():
    """Calls the C ABI without the nn.Module layer; bad arguments return error codes."""
    from opendpd_amd import _lib
    lib = _lib.load()
    assert lib.odpd_abi_version() == _lib.ABI_VERSION and lib.odpd_built_arch() == b"gfx950"
    d = _lib.ModelDesc(_lib.BACKBONE_IDS["gru"], 11, 0.0, 0.0, 0, 0, 0)
    P = lib.odpd_param_count(C.byref(d))
    assert P == 519
    B, T = 4, 16
    params = torch.randn(P, device="cuda") * 0.3
    x = torch.rand(B, T, 2, device="cuda") + 0.1
    y = torch.empty_like(x)
    assert lib.odpd_backbone_fwd(_lib.stream_ptr(), C.byref(d), B, T, _lib.ptr(params), _lib.ptr(x), _lib.ptr(y), None,
                                 None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    assert lib.odpd_backbone_fwd(_lib.stream_ptr(), C.byref(d), 0, T, _lib.ptr(params), _lib.ptr(x), _lib.ptr(y), None,
                                 None) == -1
    bad = _lib.ModelDesc(_lib.BACKBONE_IDS["gru"], 100, 0.0, 0.0, 0, 0, 0)
    assert lib.odpd_backbone_fwd(_lib.stream_ptr(), C.byref(bad), B, T, _lib.ptr(params), _lib.ptr(x), _lib.ptr(y),
                                 None, None) == -2


def test_loss_and_adamw_kernels_match_oracle():
    from opendpd_amd import _lib
    from oracle.oracle import Oracle
    lib = _lib.load()
    o = Oracle("f32")
    rng = np.random.RandomState(0)
    for kind in ("l2", "l1"):
        y = rng.randn(37, 50, 2).astype(np.float32)
        t = rng.randn(37, 50, 2).astype(np.float32)
        lo, dyo = o.loss(kind, y, t)
        yt, tt = torch.from_numpy(y).cuda(), torch.from_numpy(t).cuda()
        dy = torch.empty_like(yt)
        out = torch.zeros(_lib.LOSS_WS, device="cuda")
        rc = lib.odpd_loss_fwd_bwd(_lib.stream_ptr(), _lib.LOSS_IDS[kind], y.size, y.size, _lib.ptr(yt), _lib.ptr(tt),
                                   _lib.ptr(dy), _lib.ptr(out))
        assert rc == 0
        assert abs(out[0].item() - lo) < 1e-5 * max(1, abs(lo))
        assert rel_err(dy.cpu().numpy(), dyo) < 1e-6
    P = 1041
    p = rng.randn(P).astype(np.float32)
    m = np.zeros(P, np.float32)
    v = np.zeros(P, np.float32)
    pt, mt, vt = (torch.from_numpy(a.copy()).cuda() for a in (p, m, v))
    for step in (1, 2, 3):
        g = (rng.randn(P) * (30.0 if step == 2 else 0.1)).astype(np.float32)   # step 2 exceeds the clip norm
        gt = torch.from_numpy(g.copy()).cuda()
        nrm = torch.zeros(1, device="cuda")
        no = o.clip_adamw(p, g, m, v, step, 5e-4, 200.0)
        rc = lib.odpd_clip_adamw_step(_lib.stream_ptr(), P, _lib.ptr(pt), _lib.ptr(gt), _lib.ptr(mt), _lib.ptr(vt),
                                      step, 5e-4, 0.9, 0.999, 1e-8, 0.01, 200.0, _lib.ptr(nrm))
        assert rc == 0
        assert abs(nrm.item() - no) < 1e-4 * no
        assert rel_err(pt.cpu().numpy(), p) < 1e-6
        assert rel_err(mt.cpu().numpy(), m) < 1e-5
        assert rel_err(vt.cpu().numpy(), v) < 1e-5
        assert rel_err(gt.cpu().numpy(), g) < 1e-5


@pytest.mark.parametrize("name,bb", [("gru_h11", "gru"), ("dgru_h13", "dgru"), ("dgru_h23", "dgru"), ("qgru_h16", "qgru")])
def test_fused_train_step_follows_reference_trajectory(name, bb):
    """Three fused steps (fwd+loss+bwd in one launch, reduce, clip 200 + AdamW) reproduce the reference's
    losses and parameters after each step (tests/golden p1..p3, m3, v3)."""
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
        assert rel_err(got, fx.flat(f"p{s}", names)) < 2e-5, s
    assert rel_err(opt.exp_avg.cpu().numpy(), fx.flat("m3", names)) < 1e-3
    assert rel_err(opt.exp_avg_sq.cpu().numpy(), fx.flat("v3", names)) < 1e-3


@pytest.mark.parametrize("bb,H,B,T", [("dgru", 13, 256, 200), ("gru", 23, 37, 50), ("dgru", 13, 1027, 64), ("dgru", 9, 3, 333)])
def test_fused_equals_unfused_gradients(bb, H, B, T):
    """The fused launch and the fwd/loss/bwd chain of separate kernels give the same gradient and loss."""
    from opendpd_amd import CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    torch.manual_seed(1)
    net = CoreModel(2, H, 1, bb).cuda()
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.6
    x = x + 0.05 * torch.sign(x)
    t = torch.randn(B, T, 2, device="cuda", generator=g) * 0.3
    loss = torch.nn.functional.mse_loss(net(x), t)
    loss.backward()
    gref = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    lf = fused_train_step(opt, x, t, "l2", 0.0)
    assert abs(lf.item() - loss.item()) < 1e-5 * max(1.0, loss.item())
    assert rel_err(opt.grad[:-4].cpu().numpy(), gref) < 2e-5
    # L1 variant against autograd through the unfused kernels
    for p in net.parameters():
        p.grad = None
    l1 = torch.nn.functional.l1_loss(net(x), t)
    l1.backward()
    gref = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
    lf = fused_train_step(opt, x, t, "l1", 0.0)
    assert abs(lf.item() - l1.item()) < 1e-5 * max(1.0, l1.item())
    assert rel_err(opt.grad[:-4].cpu().numpy(), gref) < 2e-5


def test_fused_step_is_bit_repeatable():
    from opendpd_amd import CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    outs = []
    for _ in range(2):
        torch.manual_seed(2)
        net = CoreModel(2, 13, 1, "dgru").cuda()
        opt = FusedAdamW(net, lr=1e-3)
        g = torch.Generator(device="cuda").manual_seed(9)
        x = torch.rand(512, 200, 2, device="cuda", generator=g) * 0.8 + 0.05
        t = torch.rand(512, 200, 2, device="cuda", generator=g)
        for _s in range(3):
            fused_train_step(opt, x, t, "l2", 200.0)
        outs.append(net.backbone.flat_params().clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("bb,H,B,T,stride", [("dgru", 13, 300, 64, 1), ("gru", 23, 50, 33, 3), ("qgru", 10, 7, 200, 2), ("lstm", 14, 300, 64, 1),
                                             ("vdlstm", 13, 50, 33, 3), ("pgjanet", 11, 7, 200, 2)])
def test_fused_step_on_frames_addressed_in_place(bb, H, B, T, stride):
    """FrameBatch (frames = windows of resident streams, odpd_train_fwd_bwd_framed) == the same frames materialised as
    (B,T,2) tensors: bit-identical parameters after three steps."""
    from opendpd_amd import CoreModel
    from opendpd_amd.train_funcs import FrameBatch, FusedAdamW, fused_train_step
    g = torch.Generator(device="cuda").manual_seed(3)
    n = (B + 40) * stride + T
    xs = (torch.rand(n, 2, device="cuda", generator=g) - 0.5) * 1.6
    xs = xs + 0.05 * torch.sign(xs)
    ys = torch.randn(n, 2, device="cuda", generator=g) * 0.3
    order = torch.randperm(B + 40, device="cuda", generator=g)[:B].contiguous()
    idx = (order * stride)[:, None] + torch.arange(T, device="cuda")[None, :]
    outs = []
    for framed in (True, False):
        torch.manual_seed(2)
        net = CoreModel(2, H, 1, bb).cuda()
        opt = FusedAdamW(net, lr=1e-3)
        for _ in range(3):
            if framed:
                loss = fused_train_step(opt, FrameBatch(xs, ys, order, T, stride), None, "l2", 200.0)
            else:
                loss = fused_train_step(opt, xs[idx].contiguous(), ys[idx].contiguous(), "l2", 200.0)
        outs.append((net.backbone.flat_params().clone(), float(loss)))
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]


@pytest.mark.parametrize("bb", ["gru", "dgru", "qgru", "qgru_amp1"])
@pytest.mark.parametrize("H", [1, 5, 13, 16, 17, 23, 32])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (7, 63), (5, 64), (2, 65), (64, 50), (9, 200), (300, 200), (5, 130)])
def test_gate_parallel_train_kernel(bb, H, B, T):
    """the reference's own batch sizes run gru_gp_train_kernel (one sequence per wave, rows r / n / head / z; fc_out, loss and features with
    lane = time step; weight gradients as 4-block MFMAs; hidden 17..32 as two unit blocks per row; the gates are parked while LDS allows — not at
    (300, 200), nor for two unit blocks at 200 steps): loss and gradient
    against the oracle (L2 and L1), and against the row-rotated fused kernel (odpd_set_tuning gp_max_batch = 0) on the same batch"""
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
    try:
        for kind in ("l2", "l1"):
            d = yo - tgt
            lo = float((d * d).mean()) if kind == "l2" else float(np.abs(d).mean())
            dy = (2 * d / d.size if kind == "l2" else np.sign(d) / d.size).astype(np.float32)
            go, _ = o.backward(m, p, x, dy, need_dx=False)
            got = {}
            for gp in (-1, 0):
                lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(gp))
                opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
                loss = fused_train_step(opt, xt, tt, kind, 0.0)
                got[gp] = (float(loss), opt.grad[:-4].cpu().numpy().copy())
            assert abs(got[-1][0] - lo) < 2e-5 * max(1.0, lo) and abs(got[-1][0] - got[0][0]) < 1e-6 * max(1.0, lo)
            assert rel_err(got[-1][1], go) < GRAD_TOL and rel_err(got[-1][1], got[0][1]) < 2e-5
            # two kernels: different summation orders ((300, 200) with two unit blocks exceeds one sequence per SIMD: row-rotated by default too)
            assert T < 50 or (H > 16 and B > 256) or not np.array_equal(got[-1][1], got[0][1])
    finally:
        lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(-1))


@pytest.mark.parametrize("bb", ["gru", "dgru", "qgru", "qgru_amp1"])
@pytest.mark.parametrize("H", [1, 8, 13, 16, 17, 23, 24, 25, 32])
@pytest.mark.parametrize("B,T", [(1, 700), (3, 2560), (2, 256), (8, 257)])
def test_evaluation_kernel_matches_the_oracle(bb, H, B, T):
    """inference on a few long sequences (net_eval / run_dpd shapes; torch.no_grad(), so no checkpoints are asked for) runs the
    gate-parallel evaluation kernel (gru_eval_kernel: one sequence per wave, the four rows of the wave do r / n / head / z; hidden 17..32 as two
    unit blocks per row): against
    the oracle, and against the row-rotated forward the same call takes when gradients are enabled"""
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
    assert H == 1 or not np.array_equal(y_eval, y_train)          # two kernels: different summation order of the recurrent sums


@pytest.mark.parametrize("bb,H", [("gru", 13), ("dgru", 13), ("dgru", 23), ("lstm", 14), ("vdlstm", 13), ("pgjanet", 11)])
def test_gate_parallel_train_kernels_loop_over_sequences(bb, H):
    """more sequences than resident single-wave workgroups (1 100 frames of 300 samples, forced with gp_max_batch): a workgroup then runs
    several sequences in turn, accumulating into the same partial-gradient row — same loss and gradient as the four-sequence-per-wave path"""
    import ctypes as C
    from opendpd_amd import CoreModel, _lib
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    lib = _lib.load()
    B, T = 1100, 300
    torch.manual_seed(3)
    net = CoreModel(2, H, 1, bb).cuda()
    g = torch.Generator(device="cuda").manual_seed(11)
    x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.6
    x = x + 0.05 * torch.sign(x)
    t = torch.randn(B, T, 2, device="cuda", generator=g) * 0.3
    got = {}
    try:
        for gp in (1 << 30, 0):
            lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(gp))
            opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
            if gp:
                assert opt.has_fused(B, T)
                rows = int(lib.odpd_partial_rows(C.byref(net.backbone.desc), B, T, 1))
                assert 0 < rows < B                     # fewer workgroups than sequences
                loss = fused_train_step(opt, x, t, "l2", 0.0)
                got[gp] = (float(loss), opt.grad[:-4].cpu().numpy().copy())
            else:
                for q in net.parameters():
                    q.grad = None
                l2 = torch.nn.functional.mse_loss(net(x), t)
                l2.backward()
                got[gp] = (l2.item(), torch.cat([q.grad.reshape(-1) for q in net.parameters()]).cpu().numpy())
    finally:
        lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(-1))
    assert abs(got[1 << 30][0] - got[0][0]) < 1e-5 * max(1.0, got[0][0])
    assert rel_err(got[1 << 30][1], got[0][1]) < 2e-5


@pytest.mark.parametrize("bb,H", [("gru", 11), ("dgru", 13), ("qgru", 10), ("qgru_amp1", 16), ("dgru", 1)])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 17), (5, 64), (2, 65), (64, 50), (9, 200), (300, 130)])
@pytest.mark.parametrize("need_dx", [False, True])
def test_split_backward_on_the_gate_parallel_kernel(bb, H, B, T, need_dx):
    """autograd / cascade backward at the reference's batch sizes: with weight gradients only, the one-sequence-per-wave fused kernel runs
    with dL/dy given (its own forward, checkpoints unread); with dL/dx asked for, the row-rotated kernel fills the first rows of the
    (larger) partials buffer and the rest is zeroed.  Same gradients as with the one-sequence-per-wave kernels off, and as the oracle's."""
    import ctypes as C
    from opendpd_amd import CoreModel, _lib
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
    dy = rng.randn(B, T, 2).astype(np.float32)
    got = {}
    try:
        for gp in (-1, 0):
            lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(gp))
            for q in net.parameters():
                q.grad = None
            xt = torch.from_numpy(x).cuda().requires_grad_(need_dx)
            net(xt).backward(torch.from_numpy(dy).cuda())
            got[gp] = (torch.cat([q.grad.reshape(-1) for q in net.parameters()]).cpu().numpy(), xt.grad.cpu().numpy() if need_dx else None)
    finally:
        lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(-1))
    assert rel_err(got[-1][0], got[0][0]) < 2e-5
    if need_dx:
        assert rel_err(got[-1][1], got[0][1]) < 2e-5
    o, m = Oracle("f32"), make_model(bb, H)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    go, dxo = o.backward(m, p, x, dy, need_dx=need_dx)
    assert rel_err(got[-1][0], go) < GRAD_TOL
    if need_dx:
        assert rel_err(got[-1][1], dxo) < GRAD_TOL
