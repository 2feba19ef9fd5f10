"""gru / dgru / qgru / qgru_amp1 / lstm with 33 .. 64 hidden units (`hidden_size` is a free argument of the reference's backbones: gru.py:4-48,
dgru.py:9-74, qgru.py:9-71, qgru_amp1.py:9-76, lstm.py:4-48; arguments.py:49-60): csrc/gru_wide.hip, csrc/lstm_wide.hip — one sequence per wave,
lane = hidden unit — against
the oracle: forward (inference and record-writing), weight gradients and dL/dx together and each alone, batches beyond the grid (the
workgroups loop over sequences), frames that are not a multiple of the 64-step chunk; a train step through the fused optimiser; the registry
builds these sizes as HIP-backed modules (until r04: ATen restatements with a warning)."""
import warnings

import numpy as np
import pytest
import torch

from tests.golden_util import rel_err

pytestmark = pytest.mark.gpu
FWD_TOL, GRAD_TOL = 2e-5, 2e-4


def _data(B, T, seed):
    rng = np.random.RandomState(seed)
    amp = 0.05 + 0.85 * rng.rand(B, T, 1)
    ph = 2 * np.pi * rng.rand(B, T, 1)
    return np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32), rng.randn(B, T, 2).astype(np.float32)


def _net(bb, H, seed):
    from opendpd_amd import CoreModel
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("error")          # no "outside the HIP kernels' envelope" warning: these sizes are kernel-backed now
        net = CoreModel(2, H, 1, bb).cuda()
    assert net.backbone.native
    with torch.no_grad():      # biases are zero after init: make them count
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    return net


@pytest.mark.parametrize("bb,H", [("gru", 33), ("gru", 48), ("gru", 64), ("dgru", 40), ("dgru", 64), ("dgru", 33), ("qgru", 36), ("qgru_amp1", 50),
                                  ("lstm", 33), ("lstm", 47), ("lstm", 64), ("vdlstm", 33), ("vdlstm", 50), ("vdlstm", 64)])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (4, 64), (5, 70), (2, 200), (70, 33)])
def test_against_oracle_ragged(bb, H, B, T):
    from oracle.oracle import Oracle, make_model
    if bb == "vdlstm":
        T = max(T, 3)      # the circular pad takes the frame's own last three samples (vdlstm.py:66-74)
    net = _net(bb, H, H * 1000 + B * 10 + T)
    x, dy = _data(B, T, B * 7 + T)
    o = Oracle("f32")
    m = make_model(bb, H)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    assert o.param_count(m) == p.size
    yo, _ = o.forward(m, p, x)
    go, dxo = o.backward(m, p, x, dy)
    with torch.no_grad():                                   # inference: no records written
        assert rel_err(net(torch.from_numpy(x).cuda()).cpu().numpy(), yo) < FWD_TOL
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    y = net(xt)
    y.backward(torch.from_numpy(dy).cuda())
    g = np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in net.parameters()])
    assert rel_err(y.detach().cpu().numpy(), yo) < FWD_TOL
    off = 0
    for k, q in net.named_parameters():                     # every parameter tensor on its own scale
        n = q.numel()
        assert rel_err(g[off:off + n], go[off:off + n]) < GRAD_TOL, k
        off += n
    assert rel_err(xt.grad.cpu().numpy(), dxo) < GRAD_TOL
    # weights alone (the trained model of a train_pa step) and dL/dx alone (the frozen PA of a cascade)
    for q in net.parameters():
        q.grad = None
    net(torch.from_numpy(x).cuda()).backward(torch.from_numpy(dy).cuda())
    g2 = np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in net.parameters()])
    assert rel_err(g2, go) < GRAD_TOL
    for q in net.parameters():
        q.requires_grad_(False)
    xt2 = torch.from_numpy(x).cuda().requires_grad_(True)
    net(xt2).backward(torch.from_numpy(dy).cuda())
    assert rel_err(xt2.grad.cpu().numpy(), dxo) < GRAD_TOL


@pytest.mark.parametrize("bb,H", [("gru", 40), ("dgru", 48), ("lstm", 40), ("vdlstm", 36)])
def test_more_sequences_than_workgroups(bb, H):
    """B beyond 4 x CUs: every workgroup walks several sequences, its row of partial gradients accumulates over them"""
    from oracle.oracle import Oracle, make_model
    B, T = 1100, 9
    net = _net(bb, H, 5)
    x, dy = _data(B, T, 11)
    o = Oracle("f32")
    m = make_model(bb, H)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    yo, _ = o.forward(m, p, x)
    go, dxo = o.backward(m, p, x, dy)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    y = net(xt)
    y.backward(torch.from_numpy(dy).cuda())
    g = np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in net.parameters()])
    assert rel_err(y.detach().cpu().numpy(), yo) < FWD_TOL and rel_err(g, go) < GRAD_TOL and rel_err(xt.grad.cpu().numpy(), dxo) < GRAD_TOL


@pytest.mark.parametrize("bb,H", [("gru", 40), ("dgru", 64), ("lstm", 50), ("vdlstm", 40)])
def test_train_steps_follow_the_oracle(bb, H):
    """three clip + AdamW steps through the fused optimiser (forward with records, loss, backward, reduction, one-workgroup optimiser step)"""
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    net = _net(bb, H, 9)
    x, t = _data(16, 50, 2)
    t = (0.3 * t).astype(np.float32)
    o = Oracle("f64")
    m = make_model(bb, H)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()]).astype(np.float64)
    sizes = [q.numel() for q in net.parameters()]
    mom, var = np.zeros_like(p), np.zeros_like(p)
    opt = FusedAdamW(net, lr=1e-3)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    for s in range(1, 4):
        loss = fused_train_step(opt, xd, td, "l2", 200.0)
        yo, _ = o.forward(m, p, x.astype(np.float64))
        lo, dyo = o.loss("l2", yo, t.astype(np.float64))
        go, _ = o.backward(m, p, x.astype(np.float64), dyo, need_dx=False)
        o.clip_adamw(p, np.ascontiguousarray(go, dtype=np.float64), mom, var, s, 1e-3, 200.0, tensor_sizes=sizes)      # in place
        assert abs(loss.item() - lo) < 2e-5 * max(1.0, lo)
        got = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
        assert rel_err(got, p) < 2e-5, s


def test_the_api_trains_a_wide_model_on_the_kernels(tmp_path):
    import os
    import pandas as pd
    import opendpd_amd as od
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    d = dict(np.load(os.path.join(golden, "dpa200_dataset.npz")))
    ds = tmp_path / "datasets" / "DPA_200MHz"
    ds.mkdir(parents=True)
    (ds / "spec.json").write_text(str(d.pop("spec")))
    for k, v in d.items():
        pd.DataFrame(v, columns=["I", "Q"]).to_csv(ds / f"{k}.csv", index=False)
    old, old_ds = os.getcwd(), os.environ.get("OPENDPD_DATASETS")
    os.chdir(tmp_path)
    os.environ["OPENDPD_DATASETS"] = str(tmp_path / "datasets")
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error", UserWarning)
            res = od.train_pa(dataset_name="DPA_200MHz", PA_backbone="dgru", PA_hidden_size=40, frame_length=50, batch_size=64, lr=2e-3, n_epochs=2,
                              seed=0, accelerator="cuda")
        assert res["status"] == "completed" and "_M_DGRU_H_40_" in os.path.basename(res["model_path"])
        hist = pd.read_csv(os.path.join("log", "DPA_200MHz", "train_pa", "history", os.path.basename(res["log_path"])))
        assert len(hist) == 2 and np.isfinite(hist["TRAIN_LOSS"]).all() and hist["TRAIN_LOSS"][1] < hist["TRAIN_LOSS"][0]
    finally:
        os.chdir(old)
        if old_ds is not None:
            os.environ["OPENDPD_DATASETS"] = old_ds
        else:
            os.environ.pop("OPENDPD_DATASETS", None)


@pytest.mark.parametrize("name", ["wide_dgru_h40", "wide_qgru_amp1_h34", "wide_gru_h48", "wide_dgru_h64", "wide_lstm_h40", "wide_vdlstm_h36",
                                  "wide_deltagru_h34", "wide_tres_h33", "wide_pgjanet_h18"])
def test_reference_fixtures_of_wide_models(name):
    """vectors produced by RUNNING the reference at these hidden sizes (oracle/gen_golden.py wide; until r04 they pinned the ATen restatements
    only): outputs, loss, every parameter's gradient, dL/dx and one clip + AdamW step on the kernels"""
    from opendpd_amd import CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from tests.golden_util import Fixture
    fx = Fixture(name)
    m = fx.meta
    assert m["num_layers"] == 1
    net = CoreModel(2, m["hidden"], 1, m["backbone"], thx=m["thx"], thh=m["thh"])
    assert net.backbone.native
    net.load_state_dict({k: torch.from_numpy(fx["sd/" + k]) for k in fx.keys("sd")})
    net = net.cuda()
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_(True)
    t = torch.from_numpy(fx["tgt"]).cuda()
    if hasattr(net.backbone, "set_debug"):
        net.backbone.set_debug(1)
    y = net(x)
    assert rel_err(y.detach().cpu().numpy(), fx["y"]) < 2e-5
    if "stats" in fx.d:      # the delta backbones' sparsity counters of this forward pass: exact
        st = net.backbone.statistics
        assert np.array_equal(np.array([st["num_dx_zeros"], st["num_dx_numel"], st["num_dh_zeros"], st["num_dh_numel"]]), fx["stats"])
    loss = torch.nn.functional.mse_loss(y, t)
    assert abs(loss.item() - fx["losses"][0]) < 1e-5 * max(1.0, fx["losses"][0])
    loss.backward()
    for k, p in net.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), fx["g/" + k]) < 3e-4, k
    assert rel_err(x.grad.cpu().numpy(), fx["gx"]) < 3e-4
    opt = FusedAdamW(net, lr=m["lr"])
    fused_train_step(opt, x.detach(), t, "l2", m["clip"])
    for k, p in net.named_parameters():
        assert rel_err(p.detach().cpu().numpy(), fx["p1/" + k]) < 3e-5, k


def test_train_dpd_with_a_wide_dpd_in_front_of_a_two_layer_pa(tmp_path):
    """both models of a cascade on the lane-per-unit kernels: train_pa of a two-layer gru, then train_dpd of a 40-unit dgru in front of it
    (DPD forward with records, frozen-PA forward with records, loss, PA backward for dL/du only, DPD backward, fused optimiser), then run_dpd —
    no ATen fallback anywhere (warnings are errors), losses finite and falling"""
    import os
    import pandas as pd
    import opendpd_amd as od
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    d = dict(np.load(os.path.join(golden, "dpa200_dataset.npz")))
    ds = tmp_path / "datasets" / "DPA_200MHz"
    ds.mkdir(parents=True)
    (ds / "spec.json").write_text(str(d.pop("spec")))
    for k, v in d.items():
        pd.DataFrame(v, columns=["I", "Q"]).to_csv(ds / f"{k}.csv", index=False)
    old, old_ds = os.getcwd(), os.environ.get("OPENDPD_DATASETS")
    os.chdir(tmp_path)
    os.environ["OPENDPD_DATASETS"] = str(tmp_path / "datasets")
    try:
        kw = dict(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=12, PA_num_layers=2, frame_length=50, batch_size=64, lr=2e-3, seed=0,
                  accelerator="cuda")
        with warnings.catch_warnings():
            warnings.simplefilter("error", UserWarning)
            res = od.train_pa(n_epochs=2, **kw)
            assert res["status"] == "completed"
            hist = pd.read_csv(os.path.join("log", "DPA_200MHz", "train_pa", "history", os.path.basename(res["log_path"])))
            assert len(hist) == 2 and np.isfinite(hist["TRAIN_LOSS"]).all() and hist["TRAIN_LOSS"][1] < hist["TRAIN_LOSS"][0]
            res = od.train_dpd(n_epochs=2, DPD_backbone="dgru", DPD_hidden_size=40, **kw)
            assert res["status"] == "completed"
            hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
            assert len(hist) == 2 and np.isfinite(hist["TRAIN_LOSS"]).all() and hist["TRAIN_LOSS"][1] < hist["TRAIN_LOSS"][0]
            out = od.run_dpd(DPD_backbone="dgru", DPD_hidden_size=40, **kw)
            assert out["status"] == "completed" and np.isfinite(pd.read_csv(out["output_path"]).to_numpy()).all()
    finally:
        os.chdir(old)
        if old_ds is not None:
            os.environ["OPENDPD_DATASETS"] = old_ds
        else:
            os.environ.pop("OPENDPD_DATASETS", None)


@pytest.mark.parametrize("bb,H", [("deltagru", 33), ("deltagru", 50), ("deltagru", 64), ("deltagru_tcnskip", 40), ("deltagru_tcnskip", 64)])
@pytest.mark.parametrize("thx,thh", [(0.0, 0.0), (0.01, 0.05)])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (4, 64), (5, 70), (2, 200), (70, 33)])
def test_delta_backbones_against_oracle(bb, H, thx, thh, B, T):
    """deltagru / TRes-DeltaGRU of 33 .. 64 units (csrc/delta_wide.hip): outputs, exact sparsity counters, weight gradients and dL/dx"""
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H * 1000 + B * 10 + T)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        net = CoreModel(2, H, 1, bb, thx=thx, thh=thh).cuda()
    assert net.backbone.native
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    rng = np.random.RandomState(B * 7 + T)
    amp = 0.05 + 0.85 * rng.rand(B, T, 1)
    amp = 0.5 * amp + 0.5 * np.repeat(amp[:, ::4], 4, axis=1)[:, :T]      # slowly varying: kept and dropped deltas both occur
    ph = 2 * np.pi * rng.rand(B, T, 1)
    x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
    dy = rng.randn(B, T, 2).astype(np.float32)
    o = Oracle("f32")
    m = make_model(bb, H, thx, thh)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    st = np.zeros(4)
    yo = o.forward(m, p, x, stats=st)
    go, dxo = o.backward(m, p, x, dy)
    net.backbone.set_debug(1)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    y = net(xt)
    s = net.backbone.statistics
    got = np.array([s["num_dx_zeros"], s["num_dx_numel"], s["num_dh_zeros"], s["num_dh_numel"]])
    assert np.abs(got - st).max() <= (0 if thh == 0 else 2), (got, st)      # (a delta within rounding of its threshold may be taken differently)
    y.backward(torch.from_numpy(dy).cuda())
    g = np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in net.parameters()])
    flips = np.abs(got - st).max()
    tol_y, tol_g = (FWD_TOL, GRAD_TOL) if flips == 0 else (5e-2, 5e-1)
    assert rel_err(y.detach().cpu().numpy(), yo) < tol_y
    assert rel_err(g, go) < tol_g
    assert rel_err(xt.grad.cpu().numpy(), dxo) < tol_g


def test_fused_entry_points_refuse_the_lane_per_unit_models():
    """forward / backward only: every fused entry point says ODPD_EUNSUPPORTED for a model these kernels serve (never a misread parameter buffer)"""
    import ctypes as C
    from opendpd_amd import CoreModel, _lib
    lib = _lib.load()
    for bb, H in (("gru", 48), ("dgru", 40), ("lstm", 50), ("vdlstm", 36), ("deltagru", 40), ("deltagru_tcnskip", 64)):
        net = CoreModel(2, H, 1, bb).cuda()
        d = net.backbone.desc
        assert int(lib.odpd_partial_rows(C.byref(d), 64, 50, 0)) == 64 and int(lib.odpd_partial_rows(C.byref(d), 64, 50, 1)) < 0
        assert int(lib.odpd_train_workspace_floats(C.byref(d), 64, 50)) < 0
        assert int(lib.odpd_frozen_loss_rows(C.byref(d), 64, 50)) < 0
        assert int(lib.odpd_framed_train_supported_shape(C.byref(d), 64, 50)) == 0
        x = torch.rand(4, 20, 2, device="cuda")
        part = torch.empty(8, net.backbone.n_flat + _lib.LOSS_COLS, device="cuda")
        rc = lib.odpd_train_fwd_bwd(_lib.stream_ptr(), C.byref(d), 0, 4, 20, 160, _lib.ptr(net.backbone.flat_params()), _lib.ptr(x), _lib.ptr(x),
                                    _lib.ptr(part), None)
        assert rc == _lib.EUNSUPPORTED if hasattr(_lib, "EUNSUPPORTED") else rc != 0


@pytest.mark.parametrize("H", [17, 18, 24, 31, 32])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (4, 64), (5, 70), (2, 200), (70, 33), (1100, 6)])
def test_pgjanet_17_to_32_units_against_oracle(H, B, T):
    """pgjanet beyond one 16-unit tile (csrc/janet_wide.hip: lane = hidden unit, the wave's halves sharing a unit's gates): outputs, every parameter's
    gradient and dL/dx against the oracle, together and each alone"""
    from oracle.oracle import Oracle, make_model
    net = _net("pgjanet", H, H * 1000 + B * 10 + T)
    x, dy = _data(B, T, B * 7 + T)
    o = Oracle("f32")
    m = make_model("pgjanet", H)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    assert o.param_count(m) == p.size
    yo, _ = o.forward(m, p, x)
    go, dxo = o.backward(m, p, x, dy)
    with torch.no_grad():
        assert rel_err(net(torch.from_numpy(x).cuda()).cpu().numpy(), yo) < FWD_TOL
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    y = net(xt)
    y.backward(torch.from_numpy(dy).cuda())
    assert rel_err(y.detach().cpu().numpy(), yo) < FWD_TOL
    off = 0
    g = np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in net.parameters()])
    for k, q in net.named_parameters():
        n = q.numel()
        assert rel_err(g[off:off + n], go[off:off + n]) < GRAD_TOL, k
        off += n
    assert rel_err(xt.grad.cpu().numpy(), dxo) < GRAD_TOL
    for q in net.parameters():
        q.grad = None
    net(torch.from_numpy(x).cuda()).backward(torch.from_numpy(dy).cuda())
    assert rel_err(np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in net.parameters()]), go) < GRAD_TOL
    for q in net.parameters():
        q.requires_grad_(False)
    xt2 = torch.from_numpy(x).cuda().requires_grad_(True)
    net(xt2).backward(torch.from_numpy(dy).cuda())
    assert rel_err(xt2.grad.cpu().numpy(), dxo) < GRAD_TOL


@pytest.mark.parametrize("H", [33, 40, 64])
@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (4, 64), (5, 70), (2, 200), (70, 33)])
def test_deltajanet_33_to_64_units_against_oracle(H, B, T):
    """deltajanet beyond two unit tiles (csrc/deltajanet_wide.hip): outputs, exact repeat counters, every parameter's gradient and dL/dx"""
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H * 1000 + B * 10 + T)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        net = CoreModel(2, H, 1, "deltajanet").cuda()
    assert net.backbone.native
    with torch.no_grad():
        for k, p in net.named_parameters():
            if "bias" in k:
                p.uniform_(-0.3, 0.3)
    x, dy = _data(B, T, B * 7 + T)
    if T > 3:
        x[:, 2] = x[:, 1]      # an exact repeat: its feature deltas are counted
    o = Oracle("f32")
    m = make_model("deltajanet", H)
    p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
    st = np.zeros(4)
    yo = o.forward(m, p, x, stats=st)
    go, dxo = o.backward(m, p, x, dy)
    net.backbone.set_debug(1)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    y = net(xt)
    s = net.backbone.statistics
    assert np.array_equal(np.array([s["num_dx_zeros"], s["num_dx_numel"], s["num_dh_zeros"], s["num_dh_numel"]]), st)
    y.backward(torch.from_numpy(dy).cuda())
    g = np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in net.parameters()])
    assert rel_err(y.detach().cpu().numpy(), yo) < FWD_TOL and rel_err(g, go) < GRAD_TOL and rel_err(xt.grad.cpu().numpy(), dxo) < GRAD_TOL
    for q in net.parameters():
        q.requires_grad_(False)
    xt2 = torch.from_numpy(x).cuda().requires_grad_(True)
    net(xt2).backward(torch.from_numpy(dy).cuda())
    assert rel_err(xt2.grad.cpu().numpy(), dxo) < GRAD_TOL
