"""GPU parity tests of the 16-sequences-per-wave fused GRU train kernel (csrc/gru_s16.hip).  The kernel is
normally selected for batches >= 16 * 4 * CUs; here `odpd_set_tuning("s16_min_batch", 0)` forces it for every
shape so that the golden trajectories, ragged shapes (B not a multiple of 16, T not a multiple of the checkpoint
stride / staging chunk) and both occupancies are covered at sizes the split kernels and the fixtures check.

Tolerances as in test_gru_family_gpu.py (fp32; sums in a different order than ATen)."""
import numpy as np
import pytest
import torch

from tests.golden_util import Fixture, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 2], ids=["occ1", "occ2"])
def force_s16(request):
    from opendpd_amd import _lib
    lib = _lib.load()
    assert lib.odpd_set_tuning(b"s16_min_batch", 0) == 0
    assert lib.odpd_set_tuning(b"s16_occupancy", request.param) == 0
    yield request.param
    lib.odpd_set_tuning(b"s16_min_batch", -1)
    lib.odpd_set_tuning(b"s16_occupancy", 0)


def _model(fx, bb):
    from opendpd_amd import CoreModel
    net = CoreModel(2, fx.meta["hidden"], 1, bb)
    net.load_state_dict({k: torch.from_numpy(fx["sd/" + k]) for k in fx.keys("sd")})
    return net.cuda()


def test_selection_and_workspace(force_s16):
    import ctypes as C
    from opendpd_amd import CoreModel, _lib
    lib = _lib.load()
    net = CoreModel(2, 13, 1, "dgru")
    d = net.backbone.desc
    assert lib.odpd_train_workspace_floats(C.byref(d), 33, 50) == 3 * 13 * 256     # 3 wave-tasks x 13 checkpoints
    assert lib.odpd_partial_rows(C.byref(d), 33, 50, 1) == 1
    big = CoreModel(2, 23, 1, "dgru").backbone.desc                                  # hidden 17..32: two unit tiles per lane
    # (r06: hidden 17 .. 24 train on the bf16-split kernel — two-step checkpoints of 64 lanes x (float4 + float2); the query answers the
    # larger of the two kernels' layouts, so that the "s16x_train" knob can be flipped on a sized buffer)
    assert lib.odpd_train_workspace_floats(C.byref(big), 33, 50) == max(3 * 13 * 2 * 256, 3 * 25 * 384)
    big32 = CoreModel(2, 29, 1, "dgru").backbone.desc                                # hidden 25..32: the exact-fp32 kernel only
    assert lib.odpd_train_workspace_floats(C.byref(big32), 33, 50) == 3 * 13 * 2 * 256
    lib.odpd_set_tuning(b"s16_min_batch", -1)
    assert lib.odpd_train_workspace_floats(C.byref(d), 33, 50) == 0                  # small batch: LDS-resident path
    assert lib.odpd_train_workspace_floats(C.byref(d), 65536, 200) == 4096 * 50 * 256
    assert lib.odpd_set_tuning(b"no_such_knob", 1) == -1


@pytest.mark.parametrize("name,bb", [("gru_h11", "gru"), ("dgru_h13", "dgru"), ("dgru_h8", "dgru"), ("qgru_h16", "qgru"),
                                     ("qgru_h10", "qgru"), ("qgru_amp1_h10", "qgru_amp1")])
def test_s16_follows_reference_trajectory(force_s16, name, bb):
    """Three fused S16 steps reproduce the reference's losses and parameters (tests/golden p1..p3, m3, v3)."""
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    fx = Fixture(name)
    net = _model(fx, bb)
    opt = FusedAdamW(net, lr=fx.meta["lr"])
    x = torch.from_numpy(fx["x"]).cuda()
    t = torch.from_numpy(fx["tgt"]).cuda()
    assert opt.train_workspace(x.shape[0], x.shape[1], x.device) is not None      # the S16 path is the one that runs
    names = fx.keys("sd")
    for s in range(1, 4):
        loss = fused_train_step(opt, x, t, "l2", fx.meta["clip"])
        assert abs(loss.item() - fx["losses"][s - 1]) < 2e-5 * max(1.0, fx["losses"][s - 1])
        got = np.concatenate([p.detach().cpu().numpy().reshape(-1) for p in net.parameters()])
        assert rel_err(got, fx.flat(f"p{s}", names)) < 2e-5, s
    assert rel_err(opt.exp_avg.cpu().numpy(), fx.flat("m3", names)) < 1e-3
    assert rel_err(opt.exp_avg_sq.cpu().numpy(), fx.flat("v3", names)) < 1e-3


@pytest.mark.parametrize("bb,H,B,T", [("dgru", 13, 256, 200), ("gru", 11, 37, 50), ("dgru", 16, 1027, 64), ("dgru", 9, 3, 333),
                                      ("qgru", 12, 17, 5), ("qgru_amp1", 7, 1, 1), ("gru", 16, 130, 31), ("dgru", 1, 16, 4)])
def test_s16_equals_unfused_gradients(force_s16, bb, H, B, T):
    """S16 fused launch == fwd / loss / bwd chain of the row-rotated split kernels (themselves oracle-checked)."""
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
    for kind, fn in (("l2", torch.nn.functional.mse_loss), ("l1", torch.nn.functional.l1_loss)):
        for p in net.parameters():
            p.grad = None
        loss = fn(net(x), t)
        loss.backward()
        gref = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
        lf = fused_train_step(opt, x, t, kind, 0.0)
        assert abs(lf.item() - loss.item()) < 1e-5 * max(1.0, loss.item()), kind
        assert rel_err(opt.grad[:-4].cpu().numpy(), gref) < 2e-5, kind


def test_s16_against_oracle(force_s16):
    """Direct check against the CPU oracle's whole train step (fwd + loss + BPTT + clip + AdamW)."""
    from opendpd_amd import CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(3)
    B, T, H = 50, 77, 13
    net = CoreModel(2, H, 1, "dgru").cuda()
    rng = np.random.default_rng(0)
    x = (rng.uniform(0.05, 0.8, (B, T, 2)) * rng.choice([-1, 1], (B, T, 2))).astype(np.float32)
    t = rng.normal(0, 0.3, (B, T, 2)).astype(np.float32)
    orc = Oracle("f32")
    m = make_model("dgru", H)
    p = net.backbone.flat_params().detach().cpu().numpy().copy()
    ea, eas = np.zeros_like(p), np.zeros_like(p)
    opt = FusedAdamW(net, lr=1e-3)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    for s in range(1, 3):
        lo = orc.train_step(m, p, x, t, ea, eas, s, 1e-3, 200.0)
        lg = fused_train_step(opt, xd, td, "l2", 200.0)
        assert abs(lg.item() - lo) < 2e-5 * max(1.0, lo)
        assert rel_err(net.backbone.flat_params().detach().cpu().numpy(), p) < 2e-5


def test_s16_is_bit_repeatable_and_large(force_s16):
    from opendpd_amd import CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    outs = []
    for _ in range(2):
        torch.manual_seed(2)
        net = CoreModel(2, 13, 1, "dgru").cuda()
        opt = FusedAdamW(net, lr=1e-3)
        g = torch.Generator(device="cuda").manual_seed(9)
        x = torch.rand(20000, 200, 2, device="cuda", generator=g) * 0.8 + 0.05     # > one task per wave
        t = torch.rand(20000, 200, 2, device="cuda", generator=g)
        for _s in range(2):
            loss = fused_train_step(opt, x, t, "l2", 200.0)
        assert torch.isfinite(loss)
        outs.append(net.backbone.flat_params().clone())
    assert torch.equal(outs[0], outs[1])


# ---- split kernels (odpd_backbone_fwd / odpd_backbone_bwd) in the S16 mapping --------------------------------------
R1_GOLDEN = [("gru_h11", "gru"), ("dgru_h13", "dgru"), ("dgru_h8", "dgru"), ("qgru_h10", "qgru"), ("qgru_h16", "qgru"),
             ("qgru_amp1_h10", "qgru_amp1")]


@pytest.mark.parametrize("name,bb", R1_GOLDEN)
def test_s16_split_kernels_golden(force_s16, name, bb):
    """forward, parameter gradients and dL/dx of the S16 split kernels against the reference vectors"""
    from tests import test_gru_family_gpu as fam
    fam.test_golden_forward_backward(name, bb)


@pytest.mark.parametrize("bb,H", [("gru", 11), ("dgru", 13), ("gru", 16), ("qgru", 10), ("qgru_amp1", 7), ("dgru", 1)])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (17, 64), (7, 65), (5, 200), (130, 63), (2, 333)])
def test_s16_split_kernels_against_oracle_ragged(force_s16, bb, H, B, T):
    from tests import test_gru_family_gpu as fam
    fam.test_against_oracle_ragged(bb, H, B, T)


@pytest.mark.parametrize("H", list(range(1, 17)))
def test_s16_split_checkpoints_carry_real_units_only_at_every_hidden_size(force_s16, H):
    """csrc/gru_s16.hip s16_ckpt_store / s16_ckpt_load (r06): full quads keep their float4, the quad with the last H % 4 units stores that many
    dwords, padding quads nothing — every split of 1 .. 16 units into (full quads, remainder) through forward + backward over several checkpoint
    blocks, against the oracle (forward, parameter gradients, dL/dx)."""
    from tests import test_gru_family_gpu as fam
    fam.test_against_oracle_ragged("dgru" if H % 2 else "gru", H, 19, 37)


def test_s16_frozen_model_gives_dx_only(force_s16):
    from tests import test_gru_family_gpu as fam
    fam.test_frozen_model_gives_dx_only()


def test_s16_cascade_follows_reference(force_s16):
    """train_dpd cascade with both models on the S16 split kernels (DPD fwd, PA fwd, loss, PA bwd dx-only, DPD bwd)"""
    from tests import test_cascade_gpu as casc
    casc.test_cascade_autograd_matches_reference("cascade_gru11_gru11", "gru", "gru")
    casc.test_cascade_fused_steps_follow_reference("cascade_gru11_gru11", "gru", "gru")


@pytest.mark.parametrize("bb,H", [("gru", 11), ("dgru", 13)])
def test_s16_split_large_batch_matches_row_rotated(bb, H):
    """default selection at B = 20000 (S16) against the row-rotated kernels forced by the tuning knob.  DGRU's
    relu(fc_hid(h)) makes the map discontinuous: among 20000 x 40 x 13 pre-activations a few lie within rounding of 0
    and flip their mask between the two arithmetic orders, so for dgru a handful of sequences may differ."""
    from opendpd_amd import CoreModel, _lib
    lib = _lib.load()
    torch.manual_seed(4)
    net = CoreModel(2, H, 1, bb).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.rand(20000, 40, 2, device="cuda", generator=g) - 0.5) * 1.6
    x = x + 0.05 * torch.sign(x)
    dy = torch.randn(20000, 40, 2, device="cuda", generator=g)
    res = []
    for min_batch in (-1, 1 << 40):
        lib.odpd_set_tuning(b"s16_min_batch", min_batch)
        for p in net.parameters():
            p.grad = None
        xt = x.clone().requires_grad_(True)
        y = net(xt)
        y.backward(dy)
        res.append((y.detach().clone(), xt.grad.clone(), torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()))
    lib.odpd_set_tuning(b"s16_min_batch", -1)
    (y0, dx0, dp0), (y1, dx1, dp1) = res
    assert rel_err(y0.cpu().numpy(), y1.cpu().numpy()) < 2e-5
    per_seq = (dx0 - dx1).abs().amax(dim=(1, 2)) / dx1.abs().max()
    flipped = int((per_seq > 2e-5).sum())
    assert flipped <= (12 if bb == "dgru" else 0), flipped
    assert rel_err(dp0.cpu().numpy(), dp1.cpu().numpy()) < (2e-3 if bb == "dgru" else 2e-4)


def test_s16_native_epoch_loop(force_s16):
    """odpd_train_epoch with the S16 kernel: frames addressed in place inside the resident streams (frame_idx path of
    the 16-sequence staging) == per-batch loop over gathered frame tensors, bit for bit."""
    from tests import test_e2e_gpu as e2e
    e2e.test_native_epoch_loop_equals_per_step_loop("dgru", 13, 50, 64)
    e2e.test_native_epoch_loop_equals_per_step_loop("gru", 11, 200, 256)


# ---- hidden 17..32: generic-tile S16 kernels (csrc/gru_s16n.hip) ------------------------------------------------------
@pytest.fixture
def force_s16n():
    from opendpd_amd import _lib
    lib = _lib.load()
    assert lib.odpd_set_tuning(b"s16_min_batch", 0) == 0
    yield
    lib.odpd_set_tuning(b"s16_min_batch", -1)


@pytest.mark.parametrize("name,bb", [("gru_h23", "gru"), ("dgru_h23", "dgru")])
def test_s16n_golden_forward_backward_and_trajectory(force_s16n, name, bb):
    from tests import test_gru_family_gpu as fam
    fam.test_golden_forward_backward(name, bb)
    if name == "dgru_h23":
        fam.test_fused_train_step_follows_reference_trajectory(name, bb)


@pytest.mark.parametrize("bb,H", [("gru", 23), ("dgru", 23), ("dgru", 32), ("qgru", 17), ("qgru_amp1", 29), ("gru", 32)])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (17, 64), (7, 65), (5, 200), (2, 333)])
def test_s16n_against_oracle_ragged(force_s16n, bb, H, B, T):
    from tests import test_gru_family_gpu as fam
    fam.test_against_oracle_ragged(bb, H, B, T)


@pytest.mark.parametrize("bb,H,B,T", [("dgru", 23, 256, 200), ("gru", 23, 37, 50), ("dgru", 32, 100, 64), ("qgru", 20, 3, 333)])
def test_s16n_fused_equals_unfused(force_s16n, bb, H, B, T):
    """fused S16N launch (L2 and L1) == autograd through the split S16N kernels (oracle-checked above)"""
    from opendpd_amd import CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    torch.manual_seed(1)
    net = CoreModel(2, H, 1, bb).cuda()
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.6
    x = x + 0.05 * torch.sign(x)
    t = torch.randn(B, T, 2, device="cuda", generator=g) * 0.3
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    assert opt.train_workspace(B, T, x.device) is not None
    for kind, fn in (("l2", torch.nn.functional.mse_loss), ("l1", torch.nn.functional.l1_loss)):
        for p in net.parameters():
            p.grad = None
        loss = fn(net(x), t)
        loss.backward()
        gref = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
        lf = fused_train_step(opt, x, t, kind, 0.0)
        assert abs(lf.item() - loss.item()) < 1e-5 * max(1.0, loss.item()), kind
        assert rel_err(opt.grad[:-4].cpu().numpy(), gref) < 2e-5, kind


def test_s16n_cascades_follow_reference(force_s16n):
    """BASELINE config 3: TRes-DeltaGRU DPD -> frozen DGRU H23 PA, and DGRU13 -> DGRU23, with the PA on the S16N kernels"""
    from tests import test_cascade_gpu as casc
    for name, d, p in (("cascade_dgru13_dgru23", "dgru", "dgru"), ("cascade_tres15_dgru23", "deltagru_tcnskip", "dgru")):
        casc.test_cascade_autograd_matches_reference(name, d, p)
        casc.test_cascade_fused_steps_follow_reference(name, d, p)


def test_knob_change_between_steps_resizes_cached_buffers():
    """The partial-row count and the workspace layout of a (B,T) shape depend on the kernel-selection knobs: an optimiser that has
    stepped once must not keep the buffers of the previous selection (a larger grid would write past them).  Two steps with a knob
    flip in between == the same two steps on fresh optimisers, bit for bit."""
    from opendpd_amd import CoreModel, _lib
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(3)
    B, T = 3000, 40
    x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.6
    x = x + 0.05 * torch.sign(x)
    t = torch.randn(B, T, 2, device="cuda", generator=g) * 0.3

    def run(reuse):
        torch.manual_seed(5)
        net = CoreModel(2, 13, 1, "dgru").cuda()
        opt = FusedAdamW(net, lr=1e-3)
        losses, shapes = [], []
        for i, mb in enumerate((-1, 0)):                 # row-rotated kernel (LDS checkpoints), then S16 (HBM workspace)
            lib.odpd_set_tuning(b"s16_min_batch", mb)
            if not reuse and i:
                fresh = FusedAdamW(net, lr=1e-3)
                fresh._ensure(x.device)
                fresh.exp_avg.copy_(opt.exp_avg); fresh.exp_avg_sq.copy_(opt.exp_avg_sq)
                fresh.step_count = opt.step_count
                opt = fresh
            losses.append(float(fused_train_step(opt, x, t, "l2", 200.0)))
            ws = opt.train_workspace(B, T, x.device)
            shapes.append((opt.partials(B, T, x.device).shape[0], 0 if ws is None else ws.numel()))
        lib.odpd_set_tuning(b"s16_min_batch", -1)
        return losses, shapes, net.backbone.flat_params().clone()
    try:
        l1, s1, p1 = run(reuse=True)
        l2, s2, p2 = run(reuse=False)
    finally:
        lib.odpd_set_tuning(b"s16_min_batch", -1)
    assert s1 == s2 and s1[0] != s1[1], (s1, s2)         # the two selections really need different buffers
    assert l1 == l2 and torch.equal(p1, p2)
