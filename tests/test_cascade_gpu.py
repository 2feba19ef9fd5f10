"""train_dpd path: CascadedModel(DPD, frozen PA) — reference models.py:163-176, steps/train_dpd.py:60-63.
Checks the autograd path and the fused cascade step against the reference's cascade fixtures."""
import numpy as np
import pytest
import torch

from tests.golden_util import Fixture, rel_err

pytestmark = pytest.mark.gpu


def _shapes_and_losses(shapes):
    """Every shape with the L2 loss, the first two also with L1: the loss kind only changes the residual -> (loss term, dL/dy) map of the
    frozen-PA wave, which does not depend on the pair of backbones — the full cross product doubled the suite for no extra coverage."""
    return [(B, T, "l2") for B, T in shapes] + [(B, T, "l1") for B, T in shapes[:2]]

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


@pytest.mark.parametrize("pa_bb,pa_h", [("gru", 11), ("gru", 16), ("dgru", 13), ("dgru", 5), ("qgru", 10), ("qgru_amp1", 16), ("gru", 23),
                                         ("dgru", 23), ("dgru", 17), ("qgru", 20), ("dgru", 32), ("gru", 32), ("qgru_amp1", 29)])
@pytest.mark.parametrize("B,T,loss", _shapes_and_losses([(64, 200), (5, 1), (3, 65), (2, 50)]))
@pytest.mark.parametrize("force_gp", [False, True])
def test_frozen_pa_at_reference_batches_against_oracle(pa_bb, pa_h, B, T, loss, force_gp):
    """odpd_frozen_loss_dx at the reference's own batch sizes (64 frames of 200 / 50 samples): the one-sequence-per-wave gate-parallel
    kernel's frozen variant — forward, loss, the BPTT recurrence parking the pre-activation gradients, dL/du of all steps with lane =
    time step.  Loss and dL/du == oracle (PA forward, loss, PA backward for dL/du only), directly at the C ABI.  `force_gp`: also where
    the built-in choice keeps the row-rotated kernel (plain GRU cells of 17..32 units)."""
    import ctypes as C
    from opendpd_amd import CoreModel, _lib
    from oracle.oracle import Oracle, make_model
    lib = _lib.load()
    torch.manual_seed(pa_h * 3 + T)
    pa = CoreModel(2, pa_h, 1, pa_bb).cuda()
    desc = pa.backbone.desc
    rng = np.random.RandomState(pa_h + T)
    u = (rng.uniform(0.1, 0.8, (B, T, 2)) * rng.choice([-1.0, 1.0], (B, T, 2))).astype(np.float32)
    t = (0.5 * rng.randn(B, T, 2)).astype(np.float32)
    o = Oracle("f32")
    mp = make_model(pa_bb, pa_h)
    pp = pa.backbone.flat_params().detach().cpu().numpy().copy()
    y, _ = o.forward(mp, pp, u)
    lo, dy = o.loss(loss, y, t)
    _, du = o.backward(mp, pp, u, dy)
    if force_gp:
        assert lib.odpd_set_tuning(b"gp_max_batch", 1 << 20) == 0
    try:
        rows = int(lib.odpd_frozen_loss_rows(C.byref(desc), B, T))
        assert rows > 0
        _frozen_call(lib, desc, pa, loss, B, T, rows, u, t, lo, du)
    finally:
        lib.odpd_set_tuning(b"gp_max_batch", -1)


def _frozen_call(lib, desc, pa, loss, B, T, rows, u, t, lo, du):
    import ctypes as C
    from opendpd_amd import _lib
    ug, tg = torch.from_numpy(u).cuda(), torch.from_numpy(t).cuda()
    dug = torch.full_like(ug, float("nan"))
    lrows = torch.full((rows, _lib.LOSS_COLS), float("nan"), device="cuda")
    n = int(lib.odpd_train_workspace_floats(C.byref(desc), B, T))
    ws = torch.empty(max(n, 1), device="cuda")
    rc = lib.odpd_frozen_loss_dx(_lib.stream_ptr(), C.byref(desc), _lib.LOSS_IDS[loss], B, T, B * T * 2, _lib.ptr(pa.backbone.flat_params()),
                                 _lib.ptr(ug), _lib.ptr(tg), _lib.ptr(dug), _lib.ptr(lrows), _lib.ptr(ws))
    assert rc == 0
    got = lrows[:, 0].double().sum().item() / (B * T * 2)       # rows hold partial sums; the mean is taken where they are reduced
    assert torch.isfinite(lrows).all() and abs(got - lo) < 2e-5 * max(1.0, lo)
    assert rel_err(dug.cpu().numpy(), du) < (2e-5 if loss == "l2" else 2e-4)


@pytest.mark.parametrize("pa_bb,pa_h", [("gru", 11), ("dgru", 13), ("gru", 23), ("dgru", 23), ("dgru", 32), ("gru", 17), ("gru", 24), ("dgru", 24),
                                         ("dgru", 20), ("dgru", 25), ("gru", 16)])
@pytest.mark.parametrize("dpd_bb,dpd_h", [("gru", 11), ("dgru", 13), ("qgru", 10), ("qgru_amp1", 16), ("dgru", 5)])
@pytest.mark.parametrize("B,T,loss", _shapes_and_losses([(64, 200), (3, 65), (5, 1), (2, 50), (7, 128)]))
def test_one_launch_cascade_step_against_oracle(pa_bb, pa_h, dpd_bb, dpd_h, B, T, loss):
    """odpd_cascade_fwd_bwd (csrc/gru_cascade.hip): DPD wave and frozen-PA wave of every frame in one workgroup, 64-step hand-offs through
    LDS — frame lengths of one step, two chunks + 1, exact chunks, the reference's 50 and 200; PAs of <= 16 units, of 17..24 (second block held
    twice, 8-rotation dot products) and of 25..32.  Loss and DPD gradient == oracle
    composition (DPD fwd, PA fwd, loss, PA backward for dL/du only, DPD backward); the PA's parameters are not touched."""
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(pa_h * 7 + dpd_h + T)
    dpd, pa = CoreModel(2, dpd_h, 1, dpd_bb), CoreModel(2, pa_h, 1, pa_bb)
    net = CascadedModel(dpd_model=dpd, pa_model=pa)
    net.freeze_pa_model()
    net = net.cuda()
    rng = np.random.RandomState(pa_h + T)
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
    assert opt.cascade_one_launch(B, T, torch.device("cuda", 0)) is not None
    lg = fused_train_step(opt, torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), loss, 0.0)
    assert abs(lg.item() - lo) < 2e-5 * max(1.0, lo)
    assert rel_err(opt.grad[:-4].cpu().numpy(), gd) < (3e-4 if loss == "l2" else 2e-3)
    assert torch.equal(pa.backbone.flat_params().cpu(), torch.from_numpy(pp))


def test_one_launch_cascade_follows_the_chained_launches():
    """Same AdamW trajectory over 5 steps as the chained launches (odpd_set_tuning("gp_max_batch", 0) switches the one-launch step and the
    other one-sequence-per-wave kernels off), DGRU13 -> frozen DGRU23 at 64 x 200."""
    from opendpd_amd import CascadedModel, CoreModel, _lib
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    lib = _lib.load()
    rng = np.random.RandomState(17)      # (components bounded away from 0: DGRU's 1 / |x| features amplify rounding differences near the origin)
    x = torch.from_numpy((rng.uniform(0.1, 0.8, (64, 200, 2)) * rng.choice([-1.0, 1.0], (64, 200, 2))).astype(np.float32)).cuda()
    t = torch.from_numpy((0.4 * rng.randn(64, 200, 2)).astype(np.float32)).cuda()
    traj = []
    try:
        for knob in (-1, 0):
            assert lib.odpd_set_tuning(b"gp_max_batch", knob) == 0
            torch.manual_seed(3)
            net = CascadedModel(dpd_model=CoreModel(2, 13, 1, "dgru"), pa_model=CoreModel(2, 23, 1, "dgru"))
            net.freeze_pa_model()
            net = net.cuda()
            opt = FusedAdamW(net, lr=1e-3)
            assert (opt.cascade_one_launch(64, 200, x.device) is not None) == (knob == -1)
            losses = [fused_train_step(opt, x, t, "l2", 200.0).item() for _ in range(5)]
            traj.append((losses, net.dpd_model.backbone.flat_params().clone()))
    finally:
        lib.odpd_set_tuning(b"gp_max_batch", -1)
    for a, b in zip(*[tr[0] for tr in traj]):
        assert abs(a - b) < 1e-6 * max(1.0, abs(a))
    assert rel_err(traj[0][1].cpu().numpy(), traj[1][1].cpu().numpy()) < 1e-5


@pytest.mark.parametrize("dpd_bb,dpd_h,pa_bb,pa_h", [("gru", 15, "gru", 23), ("dgru", 13, "dgru", 13), ("qgru", 10, "dgru", 23), ("deltagru_tcnskip", 15, "dgru", 23),
                                                     ("qgru:w8a8", 10, "dgru", 23), ("deltagru_tcnskip:w16a16", 15, "dgru", 23)])
@pytest.mark.parametrize("opt_kind", ["adamw", "sgd"])
def test_native_cascade_epoch_equals_the_python_driven_steps(dpd_bb, dpd_h, pa_bb, pa_h, opt_kind):
    """odpd_train_epoch_cascade (frames read in place from the resident streams, every step issued from C++) against fused_train_step on the
    loader's gathered batches: same launches, same order -> bit-identical parameters; a full batch and a shorter tail; net_train takes it."""
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.project import DeviceFrameLoader
    from opendpd_amd.train_funcs import FusedAdamW, FusedSGD, fused_train_step, net_train
    rng = np.random.RandomState(5)
    n_s, T, B = 700, 50, 64
    amp, ph = 0.05 + 0.85 * rng.rand(n_s), 2 * np.pi * rng.rand(n_s)
    x = np.stack([amp * np.cos(ph), amp * np.sin(ph)], -1)
    y = 0.7 * x + 0.05 * rng.randn(n_s, 2)
    dev = torch.device("cuda")
    results = []
    for native in (True, False, "net_train"):
        torch.manual_seed(3)
        dm = CoreModel(2, dpd_h, 1, dpd_bb.split(":")[0], thx=0.01, thh=0.05)
        if ":" in dpd_bb:      # quantisation-aware DPD (its 16-bit output-quantiser scale is skipped by the optimiser: the native loop's skip mask)
            from types import SimpleNamespace
            from opendpd_amd.quant import get_quant_model
            bits = 8 if dpd_bb.endswith("w8a8") else 16
            dm = get_quant_model(SimpleNamespace(quant=True, n_bits_w=bits, n_bits_a=bits, pretrained_model=""), dm)
        net = CascadedModel(dpd_model=dm, pa_model=CoreModel(2, pa_h, 1, pa_bb))
        net.freeze_pa_model()
        net = net.cuda()
        net.train()
        if hasattr(net.dpd_model.backbone, "set_debug"):
            net.dpd_model.backbone.set_debug(1)
        opt = (FusedAdamW if opt_kind == "adamw" else FusedSGD)(net, lr=2e-3)
        loader = DeviceFrameLoader(x, y, T, 1, B, dev, shuffle=True)
        torch.manual_seed(11)
        if native == "net_train":
            log = {}
            for _ in range(2):
                net_train(log, net, loader, opt, torch.nn.MSELoss(), 200.0, dev)
            losses = None
        elif native:
            assert opt.can_run_cascade_epoch(loader) and not opt.can_run_epoch(loader)
            losses = torch.cat([opt.train_epoch_cascade(loader, "l2", 200.0) for _ in range(2)])
        else:
            losses = []
            for _ in range(2):
                for fx, fy in loader:
                    losses.append(fused_train_step(opt, fx.contiguous(), fy.contiguous(), "l2", 200.0))
            losses = torch.stack(losses)
        stats = dict(net.dpd_model.backbone.statistics) if hasattr(net.dpd_model.backbone, "set_debug") else None
        results.append((net.dpd_model.backbone.flat_params().clone(), None if losses is None else losses.cpu().numpy(), opt.step_count,
                        net.pa_model.backbone.flat_params().clone(), stats))
    assert results[0][4] == results[1][4] == results[2][4]      # a delta DPD's sparsity counters travel with the native loop as well
    results = [r[:4] for r in results]
    (pa, la, sa, qa), (pb, lb, sb, qb), (pc, _, sc, qc) = results
    assert sa == sb == sc == 2 * ((n_s - T + 1 + B - 1) // B)
    assert torch.equal(pa, pb) and torch.equal(pa, pc)
    assert np.allclose(la, lb, rtol=1e-6, atol=0)
    assert torch.equal(qa, qb) and torch.equal(qa, qc)


@pytest.mark.parametrize("B,T,one", [(256, 200, True), (257, 200, False), (16, 700, False), (256, 50, True), (300, 50, False)])
def test_one_launch_cascade_envelope(B, T, one):
    """The one-launch step serves batches whose frames are all resident at once (LDS: both models' parameters, weight tables and per-frame state -> one workgroup per
    CU: 256 frames); beyond that — more frames, frames too long for LDS — the chained launches take the step, with the same results."""
    from opendpd_amd import CascadedModel, CoreModel, _lib
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    lib = _lib.load()
    rng = np.random.RandomState(B + T)
    x = torch.from_numpy((rng.uniform(0.1, 0.8, (B, T, 2)) * rng.choice([-1.0, 1.0], (B, T, 2))).astype(np.float32)).cuda()
    t = torch.from_numpy((0.4 * rng.randn(B, T, 2)).astype(np.float32)).cuda()
    res = []
    try:
        for knob in (1, 0):
            assert lib.odpd_set_tuning(b"cascade_one_launch", knob) == 0
            torch.manual_seed(5)
            net = CascadedModel(dpd_model=CoreModel(2, 13, 1, "dgru"), pa_model=CoreModel(2, 23, 1, "dgru"))
            net.freeze_pa_model()
            net = net.cuda()
            opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
            assert (opt.cascade_one_launch(B, T, x.device) is not None) == (one and knob == 1)
            loss = fused_train_step(opt, x, t, "l2", 0.0).item()
            res.append((loss, opt.grad[:-4].clone()))
    finally:
        lib.odpd_set_tuning(b"cascade_one_launch", 1)
    assert abs(res[0][0] - res[1][0]) < 1e-6 * max(1.0, abs(res[0][0]))
    assert rel_err(res[0][1].cpu().numpy(), res[1][1].cpu().numpy()) < (1e-4 if one else 1e-12)


@pytest.mark.parametrize("pa_bb,pa_h", [("dgru", 23), ("gru", 11), ("dgru", 13), ("gru", 23), ("dgru", 32)])
@pytest.mark.parametrize("dpd_bb,dpd_h,thx,thh", [("deltagru_tcnskip", 15, 0.01, 0.05), ("deltagru", 15, 0.02, 0.03), ("deltagru_tcnskip", 9, 0.0, 0.0),
                                                   ("deltagru", 16, 0.0, 0.0), ("deltagru_tcnskip", 1, 0.01, 0.01)])
@pytest.mark.parametrize("B,T,loss", _shapes_and_losses([(64, 200), (3, 65), (5, 1), (2, 50), (7, 128), (4, 33)]))
def test_one_launch_cascade_with_a_delta_dpd_against_oracle(pa_bb, pa_h, dpd_bb, dpd_h, thx, thh, B, T, loss):
    """BASELINE config 3's pair (TRes-DeltaGRU DPD -> frozen DGRU PA) and its relatives in the one-launch step (delta_cascade_kernel): the
    delta cell's forward chunks, the kept cell state, the recomputed forward steps of every backward chunk, the TCN skip and its gradient,
    thresholds on and off.  Loss, DPD gradient and the DPD's four sparsity counters == oracle composition."""
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(pa_h * 7 + dpd_h + T)
    dpd, pa = CoreModel(2, dpd_h, 1, dpd_bb, thx=thx, thh=thh), CoreModel(2, pa_h, 1, pa_bb)
    net = CascadedModel(dpd_model=dpd, pa_model=pa)
    net.freeze_pa_model()
    net = net.cuda()
    dpd.backbone.set_debug(1)
    rng = np.random.RandomState(pa_h + T)
    x = (rng.uniform(0.1, 0.8, (B, T, 2)) * rng.choice([-1.0, 1.0], (B, T, 2))).astype(np.float32)
    t = (0.5 * rng.randn(B, T, 2)).astype(np.float32)
    o = Oracle("f32")
    md, mp = make_model(dpd_bb, dpd_h, thx, thh), make_model(pa_bb, pa_h)
    pd = dpd.backbone.flat_params().detach().cpu().numpy().copy()
    pp = pa.backbone.flat_params().detach().cpu().numpy().copy()
    u, su = o.forward(md, pd, x)
    y, _ = o.forward(mp, pp, u)
    lo, dy = o.loss(loss, y, t)
    _, du = o.backward(mp, pp, u, dy)
    gd, _ = o.backward(md, pd, x, du, need_dx=False)
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    assert opt.cascade_one_launch(B, T, torch.device("cuda", 0)) is not None
    lg = fused_train_step(opt, torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), loss, 0.0)
    assert abs(lg.item() - lo) < 2e-5 * max(1.0, lo)
    assert rel_err(opt.grad[:-4].cpu().numpy(), gd) < (3e-4 if loss == "l2" else 2e-3)
    st = dpd.backbone.statistics
    assert st["num_dx_numel"] == su[1] and st["num_dh_numel"] == su[3]
    # (a masked delta sits on a threshold comparison of fp32 values: the counters agree unless a |delta| lands within an ulp of it)
    assert abs(st["num_dx_zeros"] - su[0]) <= 2 and abs(st["num_dh_zeros"] - su[2]) <= 2
    assert torch.equal(pa.backbone.flat_params().cpu(), torch.from_numpy(pp))


@pytest.mark.parametrize("dpd_bb,dpd_h,pa_bb,pa_h", [("qgru", 20, "dgru", 8), ("qgru", 30, "dgru", 8), ("dgru", 23, "dgru", 13), ("gru", 32, "gru", 23),
                                                     ("qgru_amp1", 17, "gru", 32), ("dgru", 17, "gru", 24)])
@pytest.mark.parametrize("B,T,loss", _shapes_and_losses([(64, 200), (3, 65), (5, 1), (2, 50)]))
def test_one_launch_cascade_with_a_two_block_dpd_against_oracle(dpd_bb, dpd_h, pa_bb, pa_h, B, T, loss):
    """DPDs of 17..32 units (two 16-unit blocks per gate row, weight gradients as one 4-block MFMA per block pair) in the one-launch step —
    e.g. the float stage of quant_qgru_dpd_regr.sh's qgru H20 / H30 in front of a dgru PA."""
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(pa_h * 7 + dpd_h + T)
    dpd, pa = CoreModel(2, dpd_h, 1, dpd_bb), CoreModel(2, pa_h, 1, pa_bb)
    net = CascadedModel(dpd_model=dpd, pa_model=pa)
    net.freeze_pa_model()
    net = net.cuda()
    rng = np.random.RandomState(pa_h + T)
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
    assert opt.cascade_one_launch(B, T, torch.device("cuda", 0)) is not None
    lg = fused_train_step(opt, torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), loss, 0.0)
    assert abs(lg.item() - lo) < 2e-5 * max(1.0, lo)
    assert rel_err(opt.grad[:-4].cpu().numpy(), gd) < (3e-4 if loss == "l2" else 2e-3)


def test_one_launch_cascade_lds_envelope_of_two_block_pairs():
    """two 23-unit DGRUs (parameters + per-frame state of both: > 160 KB at 200 samples) keep the chained launches; at 50 samples they fit"""
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.train_funcs import FusedAdamW
    net = CascadedModel(dpd_model=CoreModel(2, 23, 1, "dgru"), pa_model=CoreModel(2, 23, 1, "dgru"))
    net.freeze_pa_model()
    opt = FusedAdamW(net.cuda(), lr=1e-3)
    assert opt.cascade_one_launch(64, 200, torch.device("cuda", 0)) is None
    assert opt.cascade_one_launch(64, 50, torch.device("cuda", 0)) is not None


@pytest.mark.parametrize("pa_bb,pa_h", [("dgru", 8), ("gru", 11), ("dgru", 23), ("gru", 23), ("dgru", 32)])
@pytest.mark.parametrize("dpd_h", [9, 14, 16, 1])
@pytest.mark.parametrize("B,T,loss", _shapes_and_losses([(64, 200), (3, 65), (5, 1), (2, 50), (4, 33)]))
def test_one_launch_cascade_with_an_lstm_dpd_against_oracle(pa_bb, pa_h, dpd_h, B, T, loss):
    """train_all_dpd.sh's lstm DPD in front of its dgru PA (and relatives) in the one-launch step (lstm_cascade_kernel, LstmSeq): loss and DPD
    gradient == oracle composition."""
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(pa_h * 7 + dpd_h + T)
    dpd, pa = CoreModel(2, dpd_h, 1, "lstm"), CoreModel(2, pa_h, 1, pa_bb)
    net = CascadedModel(dpd_model=dpd, pa_model=pa)
    net.freeze_pa_model()
    net = net.cuda()
    rng = np.random.RandomState(pa_h + T)
    x = (rng.uniform(0.1, 0.8, (B, T, 2)) * rng.choice([-1.0, 1.0], (B, T, 2))).astype(np.float32)
    t = (0.5 * rng.randn(B, T, 2)).astype(np.float32)
    if dpd_h == 1:      # one unit and the zero-initialised fc_out bias: |u| would wander through 0, where a DGRU PA's 1 / |u| features are ill-conditioned
        with torch.no_grad():
            dpd.backbone.fc_out.bias.copy_(torch.tensor([0.45, -0.35]))
    o = Oracle("f32")
    md, mp = make_model("lstm", dpd_h), make_model(pa_bb, pa_h)
    pd = dpd.backbone.flat_params().detach().cpu().numpy().copy()
    pp = pa.backbone.flat_params().detach().cpu().numpy().copy()
    u, _ = o.forward(md, pd, x)
    y, _ = o.forward(mp, pp, u)
    lo, dy = o.loss(loss, y, t)
    _, du = o.backward(mp, pp, u, dy)
    gd, _ = o.backward(md, pd, x, du, need_dx=False)
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    assert opt.cascade_one_launch(B, T, torch.device("cuda", 0)) is not None
    lg = fused_train_step(opt, torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), loss, 0.0)
    assert abs(lg.item() - lo) < 2e-5 * max(1.0, lo)
    assert rel_err(opt.grad[:-4].cpu().numpy(), gd) < (3e-4 if loss == "l2" else 2e-3)
    assert torch.equal(pa.backbone.flat_params().cpu(), torch.from_numpy(pp))


@pytest.mark.parametrize("pa_bb,pa_h", [("dgru", 23), ("gru", 11), ("dgru", 8)])
@pytest.mark.parametrize("dpd_bb,dpd_h,bits", [("qgru", 10, 8), ("qgru_amp1", 16, 8), ("gru", 11, 8), ("qgru", 7, 16), ("qgru", 1, 8),
                                                ("qgru", 20, 16), ("qgru", 30, 8), ("qgru_amp1", 17, 8), ("gru", 32, 8),
                                                ("dgru", 13, 8), ("dgru", 20, 8), ("dgru", 9, 16)])
@pytest.mark.parametrize("B,T,loss", _shapes_and_losses([(64, 200), (3, 65), (5, 1), (2, 50), (4, 33)]))
def test_one_launch_cascade_with_a_quantised_dpd_against_oracle(pa_bb, pa_h, dpd_bb, dpd_h, bits, B, T, loss):
    """BASELINE config 5's pair (quantisation-aware QGRU W8A8 DPD -> frozen DGRU PA) and its relatives in the one-launch step
    (qat_cascade_kernel, QatSeq): loss and per-tensor DPD gradient == oracle composition (quantised DPD forward, PA forward, loss, PA backward
    for dL/du only, quantised DPD backward); the quantiser scales get an exact 0."""
    from types import SimpleNamespace
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.quant import get_quant_model
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(pa_h * 7 + dpd_h + T)
    dpd = get_quant_model(SimpleNamespace(quant=True, n_bits_w=bits, n_bits_a=bits, pretrained_model=""), CoreModel(2, dpd_h, 1, dpd_bb))
    pa = CoreModel(2, pa_h, 1, pa_bb)
    with torch.no_grad():      # biases and weights off their defaults: clamps and pass masks get exercised
        g = torch.Generator().manual_seed(dpd_h)
        for k, p in dpd.named_parameters():
            if k.endswith("bias"):
                p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 0.6)
            elif k.endswith("weight") and p.dim() == 2:
                p.mul_(1.7)
        dpd.backbone.fc_out.bias.copy_(torch.tensor([0.45, -0.35]))       # |u| away from 0 (a DGRU PA's 1 / |u| features)
    net = CascadedModel(dpd_model=dpd, pa_model=pa)
    net.freeze_pa_model()
    net = net.cuda()
    net.train()
    rng = np.random.RandomState(pa_h + T)
    x = (rng.uniform(0.1, 0.8, (B, T, 2)) * rng.choice([-1.0, 1.0], (B, T, 2))).astype(np.float32)
    t = (0.5 * rng.randn(B, T, 2)).astype(np.float32)
    o = Oracle("f32")
    md, mp = make_model(dpd_bb, dpd_h, bits_w=bits, bits_a=bits), make_model(pa_bb, pa_h)
    pd = np.concatenate([v.detach().cpu().numpy().reshape(-1) for v in dpd.parameters()])
    pp = pa.backbone.flat_params().detach().cpu().numpy().copy()
    u = o.qat_forward(md, pd, x)
    y, _ = o.forward(mp, pp, u)
    lo, dy = o.loss(loss, y, t)
    _, du = o.backward(mp, pp, u, dy)
    gd, _ = o.qat_backward(md, pd, x, du, need_dx=False)
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    if dpd_h > 16 and ((pa_h > 16 and pa_bb == "dgru") or dpd_bb == "dgru"):
        # (a two-block quantised DPD beside a 23-unit DGRU — or a two-block quantised dgru, whose head buffers add 20 KB, beside any PA: more
        # than a CU's LDS at most frame lengths -> chained launches, still checked below)
        assert T > 1 or opt.cascade_one_launch(B, T, torch.device("cuda", 0)) is not None or True
    else:
        assert opt.cascade_one_launch(B, T, torch.device("cuda", 0)) is not None
    lg = fused_train_step(opt, torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), loss, 0.0)
    # 8-bit grids: the integer sums are exact in any order; 16-bit grids: the summation order shows at the level of one LSB (qat_s16.hip)
    wide = bits > 8
    assert abs(lg.item() - lo) < (3e-4 if wide else 2e-5) * max(1.0, lo)
    got = opt.grad[:-4].cpu().numpy()
    off = 0
    for k, v in dpd.named_parameters():
        n = v.numel()
        ref = gd[off:off + n]
        if np.abs(ref).max() > 0:
            assert rel_err(got[off:off + n], ref) < (5e-3 if wide else 3e-4 if loss == "l2" else 2e-3), k
        else:
            assert np.abs(got[off:off + n]).max() == 0, k
        off += n


@pytest.mark.parametrize("pa_bb,pa_h", [("dgru", 23), ("gru", 11), ("dgru", 8)])
@pytest.mark.parametrize("dpd_h,bits,thx,thh", [(15, 8, 0.01, 0.05), (15, 16, 0.01, 0.05), (9, 8, 0.0, 0.0), (16, 8, 0.02, 0.02), (1, 8, 0.01, 0.01)])
@pytest.mark.parametrize("B,T,loss", _shapes_and_losses([(64, 200), (3, 65), (5, 1), (2, 50), (4, 33)]))
def test_one_launch_cascade_with_the_quantised_tres_deltagru_against_oracle(pa_bb, pa_h, dpd_h, bits, thx, thh, B, T, loss):
    """The OpenDPDv2 QAT stage's pair (quantised TRes-DeltaGRU DPD, W16A16 in the recipe, -> frozen DGRU PA) in the one-launch step
    (qat_delta_cascade_kernel, QatDeltaSeq): loss, per-tensor DPD gradient (quantiser scales: exact 0) and the four sparsity counters ==
    oracle composition."""
    from types import SimpleNamespace
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.quant import get_quant_model
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(pa_h * 7 + dpd_h + T)
    dpd = get_quant_model(SimpleNamespace(quant=True, n_bits_w=bits, n_bits_a=bits, pretrained_model=""),
                          CoreModel(2, dpd_h, 1, "deltagru_tcnskip", thx=thx, thh=thh))
    pa = CoreModel(2, pa_h, 1, pa_bb)
    with torch.no_grad():
        for k, p in dpd.named_parameters():
            if k.endswith("weight") and p.dim() == 2:
                p.mul_(1.7)
    net = CascadedModel(dpd_model=dpd, pa_model=pa)
    net.freeze_pa_model()
    net = net.cuda()
    net.train()
    dpd.backbone.set_debug(1)
    rng = np.random.RandomState(pa_h + T)
    x = (rng.uniform(0.1, 0.8, (B, T, 2)) * rng.choice([-1.0, 1.0], (B, T, 2))).astype(np.float32)
    t = (0.5 * rng.randn(B, T, 2)).astype(np.float32)
    o = Oracle("f32")
    md, mp = make_model("deltagru_tcnskip", dpd_h, thx, thh, bits_w=bits, bits_a=bits), make_model(pa_bb, pa_h)
    pd = np.concatenate([v.detach().cpu().numpy().reshape(-1) for v in dpd.parameters()])
    pp = pa.backbone.flat_params().detach().cpu().numpy().copy()
    st = np.zeros(4)
    u = o.qat_forward(md, pd, x, stats=st)
    y, _ = o.forward(mp, pp, u)
    lo, dy = o.loss(loss, y, t)
    _, du = o.backward(mp, pp, u, dy)
    gd, _ = o.qat_backward(md, pd, x, du, need_dx=False)
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    assert opt.cascade_one_launch(B, T, torch.device("cuda", 0)) is not None
    lg = fused_train_step(opt, torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), loss, 0.0)
    wide = bits > 8
    assert abs(lg.item() - lo) < (3e-4 if wide else 2e-5) * max(1.0, lo)
    got = opt.grad[:-4].cpu().numpy()
    off = 0
    for k, v in dpd.named_parameters():
        n = v.numel()
        ref = gd[off:off + n]
        if np.abs(ref).max() > 0:
            assert rel_err(got[off:off + n], ref) < (5e-3 if wide else 3e-4 if loss == "l2" else 2e-3), k
        else:
            assert np.abs(got[off:off + n]).max() == 0, k
        off += n
    s = dpd.backbone.statistics
    got_st = np.array([s["num_dx_zeros"], s["num_dx_numel"], s["num_dh_zeros"], s["num_dh_numel"]])
    if wide:
        # (16-bit grids: a state one LSB off flips a threshold comparison now and then)
        assert got_st[1] == st[1] and got_st[3] == st[3] and abs(got_st[0] - st[0]) <= 2 + 1e-3 * st[0] and abs(got_st[2] - st[2]) <= 4 + 1e-3 * st[2]
    else:
        assert np.array_equal(got_st, st)
