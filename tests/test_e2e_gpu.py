"""End-to-end on the GPU: train_pa -> train_dpd -> run_dpd through the opendpd.api mirror on the bundled DPA_200MHz data
(shipped as the data fixture tests/golden/dpa200_dataset.npz), compared with what the REFERENCE logged for the same
commands (tests/golden/ref_runs.json, produced by oracle/gen_run_anchors.py running the reference on CPU).

Same seed -> same initial weights and same shuffle order, so the whole trajectory is comparable: the anchors below are
the reference's own CSV log rows.  Tolerances absorb fp32 summation-order differences accumulated over 360 (720) steps."""
import json
import os

import numpy as np
import pandas as pd
import pytest
import torch

from tests.golden_util import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    wd = tmp_path_factory.mktemp("odpd_run")
    d = dict(np.load(os.path.join(GOLDEN, "dpa200_dataset.npz")))
    ds = wd / "datasets" / "DPA_200MHz"
    ds.mkdir(parents=True)
    (ds / "spec.json").write_text(str(d.pop("spec")))
    for k, v in d.items():
        pd.DataFrame(v, columns=["I", "Q"]).to_csv(ds / f"{k}.csv", index=False)
    old = os.getcwd()
    os.chdir(wd)
    os.environ["OPENDPD_DATASETS"] = str(wd / "datasets")
    yield wd
    os.chdir(old)


def _rows_match(hist, rh, n_epochs, tol_loss, tol_db, tol_other=None):
    """every logged column but the wall-clock one: identifiers and LR exactly, *_LOSS relative, the dB metrics (and anything else, e.g. the
    sparsity columns, with `tol_other`) absolutely"""
    assert list(hist.columns) == list(rh.keys())
    for ep in range(n_epochs):
        for col in rh:
            if col == "TIME:":
                continue
            a, b = hist[col][ep], rh[col][ep]
            if isinstance(b, str) or col in ("EPOCH", "N_EPOCH", "LR", "BATCH_SIZE", "N_PARAM", "FRAME_LENGTH", "HIDDEN_SIZE", "THX", "THH"):
                assert a == b, (col, ep, a, b)
            elif col.endswith("_LOSS"):
                assert abs(a - b) < tol_loss * b + 2e-8, (col, ep, a, b)
            elif col.startswith(("VAL_", "TEST_")):
                assert abs(a - b) < tol_db, (col, ep, a, b)
            else:
                assert abs(a - b) < (tol_other if tol_other is not None else tol_db), (col, ep, a, b)


def _ref():
    return json.load(open(os.path.join(GOLDEN, "ref_runs.json")))


def test_train_pa_trajectory_matches_reference_log(workdir):
    import opendpd_amd as od
    res = od.train_pa(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, frame_length=50, batch_size=64, lr=1e-3,
                      n_epochs=2, seed=0, accelerator="cuda")
    ref = _ref()
    assert res["status"] == "completed"
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["paths"]["pa_model"])
    hist = pd.read_csv(os.path.join("log", "DPA_200MHz", "train_pa", "history", os.path.basename(res["log_path"])))
    rh = ref["train_pa_hist"]
    assert list(hist.columns) == list(rh.keys())
    for col in ("N_PARAM", "BATCH_SIZE", "FRAME_LENGTH", "HIDDEN_SIZE", "N_EPOCH"):
        assert list(hist[col]) == rh[col]
    for ep in range(2):
        assert abs(hist["TRAIN_LOSS"][ep] - rh["TRAIN_LOSS"][ep]) < 2e-3 * rh["TRAIN_LOSS"][ep]
        for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_EVM", "TEST_ACLR_AVG"):
            assert abs(hist[col][ep] - rh[col][ep]) < 0.15, (col, ep, hist[col][ep], rh[col][ep])   # dB
    sd = torch.load(res["model_path"])
    m = dict(np.load(os.path.join(GOLDEN, "ref_runs_models.npz")))
    assert list(sd.keys()) == [k[3:] for k in m if k.startswith("pa/")]


def test_mirrored_default_accelerator_runs_on_the_hip_device(workdir):
    """opendpd/api.py:35 defaults accelerator to 'cpu'; there is no CPU path here, so the mirrored default is mapped to the HIP device
    with a warning instead of failing every call that relies on it (od.train_pa(dataset_name=...), OpenDPDTrainer's implicit train_pa)."""
    import opendpd_amd as od
    with pytest.warns(UserWarning, match="no CPU path"):
        res = od.train_pa(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=8, frame_length=50, batch_size=64, n_epochs=1)
    assert res["status"] == "completed" and os.path.exists(res["model_path"])
    assert all(v.is_cuda or True for v in torch.load(res["model_path"]).values())


def test_train_dpd_and_run_dpd_match_reference(workdir):
    import opendpd_amd as od
    ref = _ref()
    m = dict(np.load(os.path.join(GOLDEN, "ref_runs_models.npz")))
    # start from the REFERENCE's trained PA so that the DPD run is comparable step by step
    os.makedirs(os.path.dirname(ref["paths"]["pa_model"]), exist_ok=True)
    torch.save({k[3:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("pa/")}, ref["paths"]["pa_model"])
    kw = dict(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, DPD_backbone="deltagru_tcnskip", DPD_hidden_size=15,
              frame_length=50, seed=0, accelerator="cuda")
    res = od.train_dpd(batch_size=64, lr=1e-3, n_epochs=1, thx=0.01, thh=0.05, **kw)
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["paths"]["dpd_model"])
    hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
    rh = ref["train_dpd_hist"]
    assert list(hist.columns) == list(rh.keys())
    assert hist["N_PARAM"][0] == rh["N_PARAM"][0] == 1518
    assert abs(hist["TRAIN_LOSS"][0] - rh["TRAIN_LOSS"][0]) < 0.03 * rh["TRAIN_LOSS"][0]
    assert abs(hist["SP_T_DX"][0] - rh["SP_T_DX"][0]) < 0.01 and abs(hist["SP_T_DH"][0] - rh["SP_T_DH"][0]) < 0.02
    for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
        assert abs(hist[col][0] - rh[col][0]) < 0.5, (col, hist[col][0], rh[col][0])   # dB, thresholded model
    # run_dpd with the REFERENCE's trained DPD weights reproduces its exported CSV
    torch.save({k[4:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("dpd/")}, ref["paths"]["dpd_model"])
    out = od.run_dpd(thx=0.01, thh=0.05, **kw)
    assert os.path.normpath(out["output_path"]) == os.path.normpath(ref["paths"]["dpd_out"])
    csv = pd.read_csv(out["output_path"])
    assert list(csv.columns) == ["I", "Q", "I_dpd", "Q_dpd"]
    assert np.abs(csv.to_numpy() - m["dpd_out"]).max() < 2e-5


@pytest.mark.parametrize("bb,H,F,B", [("dgru", 13, 50, 64), ("gru", 11, 200, 256), ("qgru", 10, 37, 100), ("bojanet", 12, 50, 64), ("apnrru", 8, 50, 64),
                                      ("dvrjanet", 12, 50, 64), ("mcldnn", 8, 50, 64), ("mcldnn", 5, 200, 256)])
def test_native_epoch_loop_equals_per_step_loop(bb, H, F, B):
    """odpd_train_epoch (frames read in place from the resident streams, C++ loop) == the Python per-batch loop over
    gathered frame tensors: same kernels, same order, bit-identical parameters and per-epoch loss.  (bojanet / apnrru / dvrjanet / mcldnn:
    their one-frame-per-workgroup fused kernels address the frames in place too.)"""
    import torch
    from opendpd_amd import CoreModel
    from opendpd_amd.project import DeviceFrameLoader
    from opendpd_amd.train_funcs import FusedAdamW, net_train
    rng = np.random.RandomState(3)
    n = 3000
    ph = np.cumsum(rng.randn(n) * 0.2)
    amp = 0.1 + 0.7 * np.abs(np.sin(np.arange(n) * 0.01 + rng.rand()))
    x = np.stack([amp * np.cos(ph), amp * np.sin(ph)], 1).astype(np.float32)
    y = (x * (1.0 - 0.3 * (x ** 2).sum(1, keepdims=True))).astype(np.float32)
    outs = []
    for native in (True, False):
        torch.manual_seed(11)
        net = CoreModel(2, H, 1, bb, **({"num_dvr_units": 3} if bb == "dvrjanet" else {})).cuda()
        opt = FusedAdamW(net, lr=2e-3)
        loader = DeviceFrameLoader(x, y, F, 1, B, torch.device("cuda"), shuffle=True)
        assert opt.can_run_epoch(loader)
        log = {}
        for _ in range(2):
            if native:
                net_train(log, net, loader, opt, torch.nn.MSELoss(), 200.0, torch.device("cuda"))
            else:
                net_train(log, net, iter(loader), opt, torch.nn.MSELoss(), 200.0, torch.device("cuda"))   # plain iterator: per-step path
        outs.append((net.backbone.flat_params().clone(), log["loss"], opt.step_count))
    assert outs[0][2] == outs[1][2] == 2 * ((n - F + 1 + B - 1) // B)
    assert torch.equal(outs[0][0], outs[1][0])
    assert abs(outs[0][1] - outs[1][1]) < 1e-6 * max(1.0, abs(outs[1][1]))


def test_gmp_train_pa_matches_reference_log(workdir):
    """the polynomial baseline (gmp.py) through train_pa on the HIP kernels — fused single-launch step inside the native epoch loop,
    long-segment evaluation (T = 2 560: five 512-sample chunks with halos) — against the rows the REFERENCE logged for the same
    command (tests/golden/ref_runs_gmp.json, oracle/gen_run_anchor_gmp.py): same seed, same initial weights and shuffle order"""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_gmp.json")))
    res = od.train_pa(dataset_name="DPA_200MHz", PA_backbone="gmp", PA_hidden_size=11, frame_length=50, batch_size=64, lr=5e-3,
                      n_epochs=2, seed=0, accelerator="cuda")
    assert res["status"] == "completed"
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["model"])
    hist = pd.read_csv(os.path.join("log", "DPA_200MHz", "train_pa", "history", os.path.basename(res["log_path"])))
    rh = ref["train_pa_hist"]
    assert list(hist.columns) == list(rh.keys())
    for col in ("N_PARAM", "BATCH_SIZE", "FRAME_LENGTH", "HIDDEN_SIZE", "N_EPOCH", "BACKBONE"):
        assert list(hist[col]) == rh[col]
    for ep in range(2):
        assert abs(hist["TRAIN_LOSS"][ep] - rh["TRAIN_LOSS"][ep]) < 2e-5 * rh["TRAIN_LOSS"][ep]      # measured 5e-7
        for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_EVM", "TEST_ACLR_AVG"):
            assert abs(hist[col][ep] - rh[col][ep]) < 1e-3, (col, ep, hist[col][ep], rh[col][ep])   # dB; measured 6e-6
    sd = torch.load(res["model_path"], map_location="cpu")
    m = dict(np.load(os.path.join(GOLDEN, "ref_runs_gmp_model.npz")))
    assert list(sd.keys()) == list(m.keys()) == ["backbone.Weight"]
    w, wr = sd["backbone.Weight"].numpy(), m["backbone.Weight"]
    assert np.abs(w - wr).max() < 1e-4 * np.abs(wr).max()      # measured 6e-7


@pytest.mark.parametrize("name,bb,H,extra", [("lstm", "lstm", 14, {}), ("tcnn", "tcnn", 35, {}), ("deltagru", "deltagru", 15, dict(thx=0.01, thh=0.05)),
                                             ("qgru", "qgru", 10, {}), ("qgru_amp1", "qgru_amp1", 10, {}), ("pgjanet", "pgjanet", 11, {})])
def test_more_backbones_follow_their_reference_logs(workdir, name, bb, H, extra):
    """two train_pa epochs on DPA_200MHz (frame 50, batch 64, lr 2e-3, seed 0) for the backbones without an anchor of their own,
    against the rows the REFERENCE logged for the same command (tests/golden/ref_runs_more.json, oracle/gen_run_anchors_more.py).
    deltagru runs with its thresholds on (thx 0.01, thh 0.05); qgru, qgru_amp1 (float) and pgjanet were run behind the harness-side
    bridge for the reference's registry defects."""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_more.json")))[name]
    res = od.train_pa(dataset_name="DPA_200MHz", PA_backbone=bb, PA_hidden_size=H, frame_length=50, batch_size=64, lr=2e-3, n_epochs=2,
                      seed=0, accelerator="cuda", **extra)
    assert res["status"] == "completed"
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["model"])
    hist = pd.read_csv(os.path.join("log", "DPA_200MHz", "train_pa", "history", os.path.basename(res["log_path"])))
    rh = ref["hist"]
    assert list(hist.columns) == list(rh.keys())
    for col in ("N_PARAM", "BATCH_SIZE", "FRAME_LENGTH", "HIDDEN_SIZE", "N_EPOCH", "BACKBONE"):
        assert list(hist[col]) == rh[col]
    for ep in range(2):       # measured: TRAIN_LOSS identical to the logged digits, metrics within 2e-4 dB (tcnn), 1e-5 dB (lstm, deltagru)
        for col in rh:            # EVERY logged column but the wall-clock one
            if col == "TIME:":
                continue
            a, b = hist[col][ep], rh[col][ep]
            if isinstance(b, str) or col in ("EPOCH", "N_EPOCH", "LR", "BATCH_SIZE", "N_PARAM", "FRAME_LENGTH", "HIDDEN_SIZE"):
                assert a == b, (col, ep, a, b)
            elif col.endswith("_LOSS"):       # the evaluation losses: mean over the evaluation batches (train_funcs.py:57-90)
                assert abs(a - b) < 2e-4 * b + 2e-8, (col, ep, a, b)
            else:
                assert abs(a - b) < 5e-3, (col, ep, a, b)   # dB


@pytest.mark.parametrize("bb", ["rvtdcnn", "bojanet", "dvrjanet", "neuraltx", "mcldnn"])
def test_f4_backbones_as_dpd_follow_their_reference_logs(workdir, bb):
    """the §8-f4 backbones in the DPD role: one train_dpd epoch (batch 64, frame 50, lr 2e-3) in front of the REFERENCE's trained gru H11 PA
    — HIP forward of the DPD, frozen-PA forward + loss + dL/du in one launch, HIP backward of the DPD, fused AdamW — against the row
    the reference logged (tests/golden/ref_runs_extras_dpd.{json,npz}, oracle/gen_run_anchors_extras_dpd.py; the reference's own
    train_dpd fails for deltajanet)"""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_extras_dpd.json")))
    m = dict(np.load(os.path.join(GOLDEN, "ref_runs_extras_dpd.npz")))
    os.makedirs(os.path.dirname(ref["pa_model"]), exist_ok=True)
    torch.save({k[3:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("pa/")}, ref["pa_model"])
    r = ref[bb]
    res = od.train_dpd(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, DPD_backbone=bb, DPD_hidden_size=r["hidden"],
                       frame_length=50, batch_size=64, lr=2e-3, n_epochs=1, seed=0, accelerator="cuda")
    assert os.path.normpath(res["model_path"]) == os.path.normpath(r["model"])
    hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
    rh = r["hist"]
    assert list(hist.columns) == list(rh.keys()) and list(hist["N_PARAM"]) == rh["N_PARAM"]
    # measured: loss 1e-7 .. 7e-6 relative, metrics 4e-6 .. 6e-3 dB (dvrjanet: its |.| kinks); bojanet, whose training is chaotic in rounding
    # (it divides by the magnitude of gain-0.1 FIR outputs): 1.2e-2 / 0.3 dB after 360 steps
    tol_l, tol_db = {"bojanet": (4e-2, 1.0), "dvrjanet": (1e-4, 3e-2)}.get(bb, (2e-5, 1e-3))
    assert abs(hist["TRAIN_LOSS"][0] - rh["TRAIN_LOSS"][0]) < tol_l * rh["TRAIN_LOSS"][0], (hist["TRAIN_LOSS"][0], rh["TRAIN_LOSS"][0])
    for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
        assert abs(hist[col][0] - rh[col][0]) < tol_db, (col, hist[col][0], rh[col][0])
    _run_dpd_with_reference_weights(od, r, m, bb, dict(DPD_backbone=bb, DPD_hidden_size=r["hidden"]))


def _run_dpd_with_reference_weights(od, r, m, bb, kw):
    """run_dpd with the REFERENCE's trained DPD weights reproduces the CSV its own run_dpd exported (where the reference's run_dpd runs)"""
    if "dpd_out" not in r:
        return
    torch.save({k[len(bb) + 4:]: torch.from_numpy(v) for k, v in m.items() if k.startswith(bb + "/sd/")}, r["model"])
    out = od.run_dpd(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, frame_length=50, seed=0, accelerator="cuda", **kw)
    assert os.path.normpath(out["output_path"]) == os.path.normpath(r["dpd_out"])
    csv = pd.read_csv(out["output_path"])
    assert list(csv.columns) == ["I", "Q", "I_dpd", "Q_dpd"]
    ref_csv = m[bb + "/dpd_out"]
    assert np.abs(csv.to_numpy() - ref_csv).max() < 2e-5 * max(1.0, np.abs(ref_csv).max())


@pytest.mark.parametrize("bb", ["gru", "dgru", "lstm", "vdlstm", "tcnn", "deltagru"])
def test_hot_path_backbones_as_dpd_follow_their_reference_logs(workdir, bb):
    """the hot-path backbones in the DPD role (TRes-DeltaGRU, GMP and the QAT cell have their own anchors above): one train_dpd epoch in
    front of the REFERENCE's trained gru H11 PA against the row the reference logged (tests/golden/ref_runs_hot_dpd.{json,npz},
    `oracle/gen_run_anchors_extras_dpd.py hot`; the reference's float qgru / qgru_amp1 cannot be built — SURVEY defect 2)"""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_hot_dpd.json")))
    m = dict(np.load(os.path.join(GOLDEN, "ref_runs_hot_dpd.npz")))
    os.makedirs(os.path.dirname(ref["pa_model"]), exist_ok=True)
    torch.save({k[3:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("pa/")}, ref["pa_model"])
    r = ref[bb]
    kw = dict(thx=0.01, thh=0.03) if bb == "deltagru" else {}
    res = od.train_dpd(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, DPD_backbone=bb, DPD_hidden_size=r["hidden"],
                       frame_length=50, batch_size=64, lr=2e-3, n_epochs=1, seed=0, accelerator="cuda", **kw)
    assert os.path.normpath(res["model_path"]) == os.path.normpath(r["model"])
    hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
    rh = r["hist"]
    assert list(hist.columns) == list(rh.keys()) and list(hist["N_PARAM"]) == rh["N_PARAM"]
    # measured: loss identical to the logged 8 digits (<= 4e-7 relative), metrics within 5e-6 dB; the thresholded deltagru 1e-7 / 4e-3 dB
    tol_l, tol_db = (1e-3, 0.05) if bb == "deltagru" else (2e-5, 1e-3)          # thresholded deltas: a decision may flip on rounding
    assert abs(hist["TRAIN_LOSS"][0] - rh["TRAIN_LOSS"][0]) < tol_l * rh["TRAIN_LOSS"][0], (hist["TRAIN_LOSS"][0], rh["TRAIN_LOSS"][0])
    for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
        assert abs(hist[col][0] - rh[col][0]) < tol_db, (col, hist[col][0], rh[col][0])
    if bb == "deltagru":
        assert abs(hist["SP_T_DX"][0] - rh["SP_T_DX"][0]) < 0.01 and abs(hist["SP_T_DH"][0] - rh["SP_T_DH"][0]) < 0.02
    _run_dpd_with_reference_weights(od, r, m, bb, dict(DPD_backbone=bb, DPD_hidden_size=r["hidden"], **kw))


def test_gmp_as_dpd_of_a_neural_pa_matches_reference(workdir):
    """the classical use: GMP pre-distorter in front of a frozen GRU PA model — train_dpd (GMP forward, frozen-PA forward + loss +
    dL/du in one launch, GMP MFMA weight gradient) and run_dpd against the reference's log row, weights and exported CSV
    (tests/golden/ref_runs_gmp.json, ref_runs_gmp_dpd.npz; oracle/gen_run_anchor_gmp.py)"""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_gmp.json")))
    m = dict(np.load(os.path.join(GOLDEN, "ref_runs_gmp_dpd.npz")))
    os.makedirs(os.path.dirname(ref["paths"]["pa_model"]), exist_ok=True)
    torch.save({k[3:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("pa/")}, ref["paths"]["pa_model"])
    kw = dict(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, DPD_backbone="gmp", DPD_hidden_size=11, frame_length=50,
              seed=0, accelerator="cuda")
    res = od.train_dpd(batch_size=64, lr=5e-3, n_epochs=1, **kw)
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["paths"]["dpd_model"])
    hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
    rh = ref["train_dpd_hist"]
    assert list(hist.columns) == list(rh.keys())
    assert hist["N_PARAM"][0] == rh["N_PARAM"][0] == 1014
    assert abs(hist["TRAIN_LOSS"][0] - rh["TRAIN_LOSS"][0]) < 2e-3 * rh["TRAIN_LOSS"][0]
    for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_EVM", "TEST_ACLR_AVG"):
        assert abs(hist[col][0] - rh[col][0]) < 0.05, (col, hist[col][0], rh[col][0])   # dB
    sd = torch.load(res["model_path"], map_location="cpu")
    w, wr = sd["backbone.Weight"].numpy(), m["dpd/backbone.Weight"]
    assert np.abs(w - wr).max() < 3e-3 * np.abs(wr).max()
    # run_dpd with the REFERENCE's trained GMP weights reproduces its exported CSV
    torch.save({k[4:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("dpd/")}, ref["paths"]["dpd_model"])
    out = od.run_dpd(**kw)
    assert os.path.normpath(out["output_path"]) == os.path.normpath(ref["paths"]["dpd_out"])
    csv = pd.read_csv(out["output_path"])
    assert list(csv.columns) == ["I", "Q", "I_dpd", "Q_dpd"]
    assert np.abs(csv.to_numpy() - m["dpd_out"]).max() < 2e-5


@pytest.mark.parametrize("name,kw", [("l1", dict(loss_type="l1", lr=1e-3)), ("clip", dict(grad_clip_val=0.02, lr=1e-3)),
                                     ("noclip", dict(grad_clip_val=0, lr=1e-3)), ("sgd", dict(opt_type="sgd", lr=1e-2)),
                                     ("adam", dict(opt_type="adam", lr=1e-3)), ("rmsprop", dict(opt_type="rmsprop", lr=1e-3))])
def test_training_options_follow_their_reference_logs(workdir, name, kw):
    """two train_pa epochs of gru H11 with the options no other anchor exercises — L1 loss, a gradient clip that really clips, no
    clipping, and the SGD(momentum 0.9) / Adam / RMSprop kinds of the fused HIP optimiser (odpd_clip_optim_step) — against the rows the
    REFERENCE logged (tests/golden/ref_runs_variants.json, oracle/gen_run_anchors_variants.py)"""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_variants.json")))[name]["hist"]
    res = od.train_pa(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, frame_length=50, batch_size=64, n_epochs=2, seed=0,
                      accelerator="cuda", **kw)
    hist = pd.read_csv(os.path.join("log", "DPA_200MHz", "train_pa", "history", os.path.basename(res["log_path"])))
    assert list(hist.columns) == list(ref.keys())
    # measured: loss 5e-6 relative, metrics 1e-5 dB (l1, whose gradient is a sign: 1.6e-4 / 2e-3 dB)
    tol_l, tol_db = (1e-3, 0.02) if name == "l1" else (5e-5, 5e-4)
    _rows_match(hist, ref, 2, tol_l, tol_db)


@pytest.mark.parametrize("name,kw", [
    ("stride7", dict(PA_backbone="dgru", PA_hidden_size=8, frame_length=37, frame_stride=7, batch_size=100, lr=2e-3)),
    ("layers2", dict(PA_backbone="gru", PA_hidden_size=8, PA_num_layers=2, frame_length=50, batch_size=64, lr=2e-3)),
    ("hidden40", dict(PA_backbone="dgru", PA_hidden_size=40, frame_length=50, batch_size=64, lr=1e-3)),
    ("lstm_layers2", dict(PA_backbone="lstm", PA_hidden_size=10, PA_num_layers=2, frame_length=50, batch_size=64, lr=2e-3)),
    ("dgru_layers2", dict(PA_backbone="dgru", PA_hidden_size=9, PA_num_layers=2, frame_length=50, batch_size=64, lr=2e-3)),
    ("gru_h48", dict(PA_backbone="gru", PA_hidden_size=48, frame_length=50, batch_size=64, lr=1e-3)),
    ("lstm_h48", dict(PA_backbone="lstm", PA_hidden_size=48, frame_length=50, batch_size=64, lr=1e-3)),
    ("vdlstm_h40", dict(PA_backbone="vdlstm", PA_hidden_size=40, frame_length=50, batch_size=64, lr=1e-3)),
    ("deltagru_h40", dict(PA_backbone="deltagru", PA_hidden_size=40, thx=0.01, thh=0.05, frame_length=50, batch_size=64, lr=1e-3))])
def test_framing_and_envelope_variants_follow_their_reference_logs(workdir, name, kw):
    """strided frames (frame_stride 7: the native epoch loop addresses frame f at f * 7 of the resident stream) and the configurations that
    left the ATen restatements in r04 — two layers (gru, lstm, dgru: csrc/*_layers2.hip) and 33 .. 64 hidden units (dgru 40, gru 48, lstm 48,
    vdlstm 40, deltagru 40 with thresholds: csrc/*_wide.hip; record-writing forward, loss, backward, fused optimiser) —, two train_pa epochs
    each, against the REFERENCE's logged rows (ref_runs_variants.json)"""
    import warnings
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_variants.json")))[name]["hist"]
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        res = od.train_pa(dataset_name="DPA_200MHz", n_epochs=2, seed=0, accelerator="cuda", **kw)
    hist = pd.read_csv(os.path.join("log", "DPA_200MHz", "train_pa", "history", os.path.basename(res["log_path"])))
    assert list(hist.columns) == list(ref.keys())
    for col in ("N_PARAM", "BATCH_SIZE", "FRAME_LENGTH", "HIDDEN_SIZE"):
        assert list(hist[col]) == ref[col]
    if name == "layers2":       # since r04 on csrc/gru_layers2.hip: no warning any more
        assert not any("outside the HIP kernels' envelope" in str(m.message) for m in caught)
    if name == "deltagru_h40":      # thresholded: 720 steps of rounding-level differences flip a few delta decisions (as config 3)
        print("[deltagru_h40] rows:", hist[["TRAIN_LOSS", "VAL_NMSE", "TEST_ACLR_AVG"]].to_numpy().tolist())
        for ep in range(2):
            assert abs(hist["TRAIN_LOSS"][ep] - ref["TRAIN_LOSS"][ep]) < 2e-3 * ref["TRAIN_LOSS"][ep]      # (measured: equal to the logged 7 digits)
            for col in ("VAL_NMSE", "TEST_NMSE", "VAL_ACLR_AVG", "TEST_ACLR_AVG"):
                assert abs(hist[col][ep] - ref[col][ep]) < 0.05, (col, ep, hist[col][ep], ref[col][ep])
    elif name in ("hidden40", "lstm_layers2", "dgru_layers2", "gru_h48", "lstm_h48", "vdlstm_h40"):
        # 720 steps on the lane-per-unit / two-layer kernels (their own sigmoid / tanh evaluations): measured 1.2e-4 on VAL_LOSS of epoch 2 (hidden40)
        _rows_match(hist, ref, 2, 6e-4, 6e-3)
    else:
        _rows_match(hist, ref, 2, 5e-5, 3e-3)       # measured: loss 5e-6 relative, metrics 1e-5 dB


@pytest.mark.parametrize("bb", ["rvtdcnn", "bojanet", "deltajanet", "dvrjanet", "neuraltx", "mcldnn"])
def test_restated_registry_backbones_follow_their_reference_logs(workdir, bb):
    """the registry names that still run as torch restatements (backbones/extras.py), one train_pa epoch through the same Project
    flow (ATen forward / backward on the GPU, torch AdamW), against the REFERENCE's logged row (tests/golden/ref_runs_extras.json,
    oracle/gen_run_anchors_extras.py; apnrru is not among the reference CLI's --PA_backbone choices)"""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_extras.json")))[bb]
    kw = dict(thx=0.01, thh=0.05) if bb == "deltajanet" else {}
    res = od.train_pa(dataset_name="DPA_200MHz", PA_backbone=bb, PA_hidden_size=ref["hidden"], frame_length=50, batch_size=256, lr=2e-3,
                      n_epochs=1, seed=0, accelerator="cuda", **kw)
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["model"])
    hist = pd.read_csv(os.path.join("log", "DPA_200MHz", "train_pa", "history", os.path.basename(res["log_path"])))
    rh = ref["hist"]
    assert list(hist.columns) == list(rh.keys()) and list(hist["N_PARAM"]) == rh["N_PARAM"]
    # measured: loss 7e-8 relative, metrics 2e-5 dB.  bojanet divides by the magnitude of gain-0.1 FIR outputs: its training amplifies
    # rounding-level differences within tens of steps (the restatement issues the reference's ATen calls in the reference's order and
    # is bit-identical with it on the SAME device over 30 steps; GPU vs the CPU-run reference: 8e-4 / 0.14 dB after 90 steps)
    tol_l, tol_db = {"bojanet": (5e-3, 0.5), "deltajanet": (1e-3, 0.05)}.get(bb, (2e-5, 1e-3))
    assert abs(hist["TRAIN_LOSS"][0] - rh["TRAIN_LOSS"][0]) < tol_l * rh["TRAIN_LOSS"][0]
    for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
        assert abs(hist[col][0] - rh[col][0]) < tol_db, (col, hist[col][0], rh[col][0])


def test_lr_schedule_run_matches_reference_log(workdir):
    """--lr_schedule 1 --patience 0 --decay_factor 0.5 --lr_end 1e-3 at lr 5e-2 (eight train_pa epochs, gru H11): ReduceLROnPlateau on
    the validation NMSE halves the rate after the third epoch in the reference's log; the LR column (logged before the scheduler
    steps, project.py:343-362) and the trajectory must follow (tests/golden/ref_runs_lrsched.json, oracle/gen_run_anchor_lrsched.py)."""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_lrsched.json")))["hist"]
    res = od.train_pa(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, frame_length=50, batch_size=64, lr=5e-2, lr_schedule=1,
                      patience=0, decay_factor=0.5, lr_end=1e-3, n_epochs=8, seed=0, accelerator="cuda")
    hist = pd.read_csv(os.path.join("log", "DPA_200MHz", "train_pa", "history", os.path.basename(res["log_path"])))
    assert list(hist.columns) == list(ref.keys())
    assert list(hist["LR"]) == ref["LR"] and ref["LR"][2] == 0.05 and ref["LR"][3] == 0.025
    _rows_match(hist, ref, 8, 2e-4, 2e-3)       # measured: loss equal to the logged digits, NMSE within 7e-5 dB over all eight epochs


def test_thresholded_dpd_two_epochs_match_reference_incl_sparsity_columns(workdir):
    """train_dpd of the TRes-DeltaGRU (thx 0.01, thh 0.05) for TWO epochs: the temporal-sparsity columns come from counters that
    the training AND the evaluation forwards feed and that are read (and reset) once per epoch (paths.py:49-59) — bookkeeping a
    one-epoch run cannot pin.  Anchor: tests/golden/ref_runs_delta2.json (oracle/gen_run_anchor_delta2.py)."""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_delta2.json")))
    pa = dict(np.load(os.path.join(GOLDEN, "ref_runs_delta2_pa.npz")))
    os.makedirs(os.path.dirname(ref["pa_model"]), exist_ok=True)
    torch.save({k: torch.from_numpy(v) for k, v in pa.items()}, ref["pa_model"])
    res = od.train_dpd(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, DPD_backbone="deltagru_tcnskip", DPD_hidden_size=15,
                       frame_length=50, seed=0, accelerator="cuda", batch_size=64, lr=1e-3, n_epochs=2, thx=0.01, thh=0.05)
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["dpd_model"])
    hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
    rh = ref["hist"]
    assert list(hist.columns) == list(rh.keys())
    for ep in range(2):       # measured: loss equal to the logged digits, sparsity 2e-6, HW_PARAM 1e-3, NMSE 2e-3 dB, ACLR 0.09 dB
        assert abs(hist["TRAIN_LOSS"][ep] - rh["TRAIN_LOSS"][ep]) < 1e-3 * rh["TRAIN_LOSS"][ep], ep
        for col, tol in (("SP_T_DX", 1e-4), ("SP_T_DH", 1e-4), ("SP_T_DV", 1e-4), ("HW_PARAM", 0.05), ("VAL_NMSE", 0.05), ("TEST_NMSE", 0.05),
                         ("VAL_ACLR_AVG", 0.3), ("TEST_ACLR_AVG", 0.3)):
            assert abs(hist[col][ep] - rh[col][ep]) < tol, (col, ep, hist[col][ep], rh[col][ep])


def test_quantised_flow_two_epochs_and_run_dpd_match_reference(workdir):
    """--quant --n_bits_w 8 --n_bits_a 8 --quant_dir_label w8a8 on DPA_200MHz: two train_dpd epochs of the QAT QGRU H10 in front of
    the reference's GRU PA (the second epoch trains AFTER an evaluation pass: the eval-only output quantiser must be off again), the
    saved checkpoint (keys, never-exercised buffers) and run_dpd's exported CSV, against a reference run
    (tests/golden/ref_runs_qat_dpa.{json,npz}, oracle/gen_run_anchor_qat_dpa.py).  The float PA in the loop differs at rounding
    level and moves values across quantisation boundaries, hence the training tolerances; run_dpd with the REFERENCE's trained
    weights is the quantised cell alone: outputs on the 2^-14 grid, equal up to one LSB."""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_qat_dpa.json")))
    m = dict(np.load(os.path.join(GOLDEN, "ref_runs_qat_dpa.npz")))
    os.makedirs(os.path.dirname(ref["pa_model"]), exist_ok=True)
    torch.save({k[3:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("pa/")}, ref["pa_model"])
    kw = dict(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, DPD_backbone="qgru", DPD_hidden_size=10, frame_length=50,
              seed=0, accelerator="cuda", quant=True, n_bits_w=8, n_bits_a=8, quant_dir_label="w8a8")
    res = od.train_dpd(batch_size=64, lr=1e-3, n_epochs=2, **kw)
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["dpd_model"])
    hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
    rh = ref["hist"]
    assert list(hist.columns) == list(rh.keys()) and list(hist["N_PARAM"]) == rh["N_PARAM"]
    for ep in range(2):
        assert abs(hist["TRAIN_LOSS"][ep] - rh["TRAIN_LOSS"][ep]) < 0.02 * rh["TRAIN_LOSS"][ep], (ep, hist["TRAIN_LOSS"][ep], rh["TRAIN_LOSS"][ep])
        for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
            assert abs(hist[col][ep] - rh[col][ep]) < 0.4, (col, ep, hist[col][ep], rh[col][ep])   # dB
    sd = torch.load(res["model_path"], map_location="cpu")
    ref_sd = {k[4:]: v for k, v in m.items() if k.startswith("dpd/")}
    assert list(sd.keys()) == list(ref_sd.keys())
    for k, v in ref_sd.items():
        if "_num" in k or "pow2_scale" in k or "n_bits" in k:
            assert np.array_equal(sd[k].numpy(), v), k          # incl. the x2h / h2h out_quantizers no forward ever touches
    torch.save({k: torch.from_numpy(v) for k, v in ref_sd.items()}, ref["dpd_model"])
    out = od.run_dpd(**kw)
    assert os.path.normpath(out["output_path"]) == os.path.normpath(ref["dpd_out"])
    csv = pd.read_csv(out["output_path"])
    assert list(csv.columns) == ["I", "Q", "I_dpd", "Q_dpd"]
    assert np.abs(csv.to_numpy() - m["dpd_out"]).max() <= 2.0 ** -14 + 1e-9


def test_quantised_head_lstm_flow_matches_reference(workdir):
    """--quant with an lstm DPD: the surgery swaps only fc_out (nn.LSTM stays float, quant_envs.py:40-60).  Two train_dpd epochs in front of
    the reference's GRU PA, the checkpoint (keys incl. the quantisers' side-effect buffers, file name with the 3 scale parameters counted)
    and run_dpd's CSV with the REFERENCE's trained weights, against tests/golden/ref_runs_qat_lstm.{json,npz}
    (oracle/gen_run_anchor_qat_lstm.py).  Tolerances as in the qgru flow; run_dpd: outputs on the 2^-14 grid, a float state next to a
    rounding boundary of the 8-bit activation grid may move single samples by a weight x 2^-6 product."""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_qat_lstm.json")))
    m = dict(np.load(os.path.join(GOLDEN, "ref_runs_qat_lstm.npz")))
    pa = dict(np.load(os.path.join(GOLDEN, "ref_runs_qat_dpa.npz")))
    os.makedirs(os.path.dirname(ref["pa_model"]), exist_ok=True)
    torch.save({k[3:]: torch.from_numpy(v) for k, v in pa.items() if k.startswith("pa/")}, ref["pa_model"])
    kw = dict(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, DPD_backbone="lstm", DPD_hidden_size=12, frame_length=50,
              seed=0, accelerator="cuda", quant=True, n_bits_w=8, n_bits_a=8, quant_dir_label="w8a8")
    res = od.train_dpd(batch_size=64, lr=1e-3, n_epochs=2, **kw)
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["dpd_model"])
    hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
    rh = ref["hist"]
    assert list(hist.columns) == list(rh.keys()) and list(hist["N_PARAM"]) == rh["N_PARAM"]
    for ep in range(2):
        assert abs(hist["TRAIN_LOSS"][ep] - rh["TRAIN_LOSS"][ep]) < 0.02 * rh["TRAIN_LOSS"][ep], (ep, hist["TRAIN_LOSS"][ep], rh["TRAIN_LOSS"][ep])
        for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
            assert abs(hist[col][ep] - rh[col][ep]) < 0.4, (col, ep, hist[col][ep], rh[col][ep])   # dB
    sd = torch.load(res["model_path"], map_location="cpu")
    ref_sd = {k[4:]: v for k, v in m.items() if k.startswith("dpd/")}
    assert list(sd.keys()) == list(ref_sd.keys())
    for k, v in ref_sd.items():
        if "_num" in k or "pow2_scale" in k or "n_bits" in k:
            assert np.array_equal(sd[k].numpy(), v), k
    torch.save({k: torch.from_numpy(v) for k, v in ref_sd.items()}, ref["dpd_model"])
    out = od.run_dpd(**kw)
    assert os.path.normpath(out["output_path"]) == os.path.normpath(ref["dpd_out"])
    csv = pd.read_csv(out["output_path"])
    assert list(csv.columns) == ["I", "Q", "I_dpd", "Q_dpd"]
    d = np.abs(csv.to_numpy() - m["dpd_out"])
    assert (d > 2.0 ** -14 + 1e-9).sum() <= 4 and d.max() < 2.0 ** -5, ((d > 2.0 ** -14 + 1e-9).sum(), d.max())


@pytest.mark.parametrize("bb", ["neuraltx", "rvtdcnn", "pgjanet"])
def test_quantised_head_neuraltx_flow_matches_reference(workdir, bb):
    """--quant with a neuraltx DPD: the surgery's layer map holds nn.Conv2d and nn.Linear (quant_envs.py:145-148), so only IQ_match changes
    (bias-free INT_Linear; no module named fc_out: no output quantiser in eval); with an rvtdcnn DPD every layer is in the map (INT_Conv2D,
    two INT_Linear, fc_out with the 16-bit output grid in eval; csrc/rvtdcnn_q.hip); with a pgjanet DPD the cell's six Linears (csrc/pgjanet_q.hip;
    the reference reaches this backbone only through a harness-side bridge of its constructor defect).  Two train_dpd epochs in front of the reference's GRU PA, the
    checkpoint (keys incl. the quantisers' side-effect buffers, file name with the 3 scale parameters counted) and run_dpd's CSV with the
    REFERENCE's trained weights, against tests/golden/ref_runs_qat_{neuraltx,rvtdcnn,pgjanet}.{json,npz} (oracle/gen_run_anchor_qat_lstm.py <backbone>)."""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, f"ref_runs_qat_{bb}.json")))
    m = dict(np.load(os.path.join(GOLDEN, f"ref_runs_qat_{bb}.npz")))
    pa = dict(np.load(os.path.join(GOLDEN, "ref_runs_qat_dpa.npz")))
    os.makedirs(os.path.dirname(ref["pa_model"]), exist_ok=True)
    torch.save({k[3:]: torch.from_numpy(v) for k, v in pa.items() if k.startswith("pa/")}, ref["pa_model"])
    kw = dict(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, DPD_backbone=bb, DPD_hidden_size=12, frame_length=50,
              seed=0, accelerator="cuda", quant=True, n_bits_w=8, n_bits_a=8, quant_dir_label="w8a8")
    res = od.train_dpd(batch_size=64, lr=1e-3, n_epochs=2, **kw)
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["dpd_model"])
    hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
    rh = ref["hist"]
    assert list(hist.columns) == list(rh.keys()) and list(hist["N_PARAM"]) == rh["N_PARAM"]
    gaps = {}
    for ep in range(2):
        gaps[f"loss{ep}"] = abs(hist["TRAIN_LOSS"][ep] - rh["TRAIN_LOSS"][ep]) / rh["TRAIN_LOSS"][ep]
        # (rvtdcnn: three 8-bit activation grids inside the trained model — a value next to a rounding boundary lands on the other side somewhere
        # in every step and the trajectories part at that level, as in the qgru / lstm flows: their tolerances)
        assert gaps[f"loss{ep}"] < (1e-3 if bb == "neuraltx" else 0.02), (ep, hist["TRAIN_LOSS"][ep], rh["TRAIN_LOSS"][ep])
        for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
            gaps[f"{col}{ep}"] = abs(hist[col][ep] - rh[col][ep])
            # (rvtdcnn measured: <= 0.38 dB after the first epoch — the loss still falls by 10 x per epoch there —, <= 0.34 dB after the second)
            assert gaps[f"{col}{ep}"] < (0.1 if bb == "neuraltx" else 0.8), (col, ep, hist[col][ep], rh[col][ep])   # dB
    # measured (r04 box): loss 2e-7 / 5.9e-5 relative, metrics <= 1e-4 dB after the first epoch, <= 0.031 dB after the second
    print(f"[qat {bb}] gaps:", {k: f"{v:.1e}" for k, v in gaps.items()})
    sd = torch.load(res["model_path"], map_location="cpu")
    ref_sd = {k[4:]: v for k, v in m.items() if k.startswith("dpd/")}
    assert list(sd.keys()) == list(ref_sd.keys())
    for k, v in ref_sd.items():
        if "_num" in k or "pow2_scale" in k or "n_bits" in k:
            assert np.array_equal(sd[k].numpy(), v), k
    torch.save({k: torch.from_numpy(v) for k, v in ref_sd.items()}, ref["dpd_model"])
    out = od.run_dpd(**kw)
    assert os.path.normpath(out["output_path"]) == os.path.normpath(ref["dpd_out"])
    csv = pd.read_csv(out["output_path"])
    assert list(csv.columns) == ["I", "Q", "I_dpd", "Q_dpd"]
    d = np.abs(csv.to_numpy() - m["dpd_out"])
    # (rvtdcnn: outputs on the 2^-14 grid; a value next to a rounding boundary of one of its three 8-bit activation grids may move single samples)
    # (pgjanet: no output grid; a state next to a rounding boundary of one of the six 8-bit activation grids INSIDE the recurrence takes the
    # rest of its frame along: run_dpd exports one long sequence, so the count is of samples behind the first such flip)
    if bb == "pgjanet":
        print(f"[qat pgjanet] run_dpd: {(d > 2e-6).sum()} of {d.size} values differ, largest {d.max():.2e}")
        assert np.median(d) <= 2e-6 or d.max() < 0.1, (np.median(d), d.max())
        return
    assert (d > (2e-6 if bb == "neuraltx" else 2.0 ** -14 + 1e-9)).sum() <= 6 and d.max() < 2.0 ** -4, ((d > 2e-6).sum(), d.max())


def test_quantised_flow_with_pretrained_float_checkpoint_matches_reference(workdir):
    """train_dpd --quant --pretrained_model <float checkpoint with the float holder's key names> (the q_pretrain -> QAT hand-over,
    quant_envs.py:173-182): the checkpoint's weights go through quantisation, its biases are re-drawn by INT_Linear, the scales start at
    their defaults.  One epoch against the reference's logged row and saved state (tests/golden/ref_runs_qat_pre.{json,npz},
    oracle/gen_run_anchor_qat_pretrained.py); tolerances as in the flow without a checkpoint."""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_qat_pre.json")))
    m = dict(np.load(os.path.join(GOLDEN, "ref_runs_qat_pre.npz")))
    pa = dict(np.load(os.path.join(GOLDEN, "ref_runs_qat_dpa.npz")))
    os.makedirs(os.path.dirname(ref["pa_model"]), exist_ok=True)
    torch.save({k[3:]: torch.from_numpy(v) for k, v in pa.items() if k.startswith("pa/")}, ref["pa_model"])
    torch.save({k[4:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("pre/")}, "pygru.pt")
    res = od.train_dpd(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, DPD_backbone="qgru", DPD_hidden_size=10, frame_length=50,
                       seed=0, accelerator="cuda", quant=True, n_bits_w=8, n_bits_a=8, quant_dir_label="w8a8pre", pretrained_model="pygru.pt",
                       batch_size=64, lr=1e-3, n_epochs=1)
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["dpd_model"])
    hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
    rh = ref["hist"]
    assert list(hist.columns) == list(rh.keys()) and list(hist["N_PARAM"]) == rh["N_PARAM"]
    assert abs(hist["TRAIN_LOSS"][0] - rh["TRAIN_LOSS"][0]) < 0.02 * rh["TRAIN_LOSS"][0], (hist["TRAIN_LOSS"][0], rh["TRAIN_LOSS"][0])
    for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
        assert abs(hist[col][0] - rh[col][0]) < 0.4, (col, hist[col][0], rh[col][0])   # dB
    sd = torch.load(res["model_path"], map_location="cpu")
    ref_sd = {k[4:]: v for k, v in m.items() if k.startswith("dpd/")}
    assert list(sd.keys()) == list(ref_sd.keys())
    # 360 AdamW steps at lr 1e-3 move a weight by at most 0.36: the trained weights are still the checkpoint's, moved; the biases were
    # re-drawn (uniform +-1/sqrt(fan_in)) and zero-gradient-free, so they are NOT the checkpoint's — in the reference's file and here
    w, b = "backbone.rnn.rnn_cell_list.0.h2h.weight", "backbone.rnn.rnn_cell_list.0.x2h.bias"
    for got in (sd[w].numpy(), ref_sd[w]):
        assert np.abs(got - m["pre/" + w]).max() < 0.45
        assert np.corrcoef(got.reshape(-1), m["pre/" + w].reshape(-1))[0, 1] > 0.5
    for got in (sd[b].numpy(), ref_sd[b]):
        assert np.abs(got - m["pre/" + b]).max() > 0.45
    assert np.abs(sd[w].numpy() - ref_sd[w]).max() < 0.05          # and the two runs stay next to each other


@pytest.fixture
def tiny_workdir(tmp_path):
    """DPA_200MHz cut down to 4 000 training samples and ONE 2 560-sample validation / test segment (for flows that run op by op in ATen)"""
    d = dict(np.load(os.path.join(GOLDEN, "dpa200_dataset.npz")))
    ds = tmp_path / "datasets" / "DPA_200MHz"
    ds.mkdir(parents=True)
    (ds / "spec.json").write_text(str(d.pop("spec")))
    for k, v in d.items():
        pd.DataFrame(v[:4000 if k.startswith("train") else 2560], columns=["I", "Q"]).to_csv(ds / f"{k}.csv", index=False)
    old, old_ds = os.getcwd(), os.environ.get("OPENDPD_DATASETS")
    os.chdir(tmp_path)
    os.environ["OPENDPD_DATASETS"] = str(tmp_path / "datasets")
    yield tmp_path
    os.chdir(old)
    if old_ds is not None:
        os.environ["OPENDPD_DATASETS"] = old_ds
    else:
        os.environ.pop("OPENDPD_DATASETS", None)


@pytest.mark.parametrize("bb,H", [("dvrjanet", 20), ("apnrru", 8)])
def test_registry_backbone_without_kernels_trains_through_the_api(tiny_workdir, bb, H):
    """SURVEY §8 f4 names without a reference-logged anchor go through the same Project flow on the GPU — dvrjanet with 20 units (beyond
    the kernels' envelope of 16: backbones/extras.py, ATen forward / backward, torch.optim.AdamW) and apnrru (HIP kernels + fused AdamW;
    the reference's CLI does not list it among its --PA_backbone choices, so there is no reference log to anchor it to): device-resident
    frame loader, eval + metrics + checkpoint / log layout.  (Until r04 the ATen case was mcldnn with 20 channels: its convolutions made
    this ONE test 175 .. 390 s of the suite — MIOpen solver search, or with MIOpen bypassed the first-use load of the GEMM kernel
    libraries — for a flow check; its restatement stays pinned by tests/test_extras_cpu.py and test_mcldnn_gpu.py.)"""
    import opendpd_amd as od
    res = od.train_pa(dataset_name="DPA_200MHz", PA_backbone=bb, PA_hidden_size=H, frame_length=50, lr=2e-3, n_epochs=2, seed=0, accelerator="cuda",
                      batch_size=256)
    assert res["status"] == "completed" and os.path.exists(res["model_path"])
    hist = pd.read_csv(os.path.join("log", "DPA_200MHz", "train_pa", "history", os.path.basename(res["log_path"])))
    assert len(hist) == 2 and np.isfinite(hist["TRAIN_LOSS"]).all()
    assert hist["TRAIN_LOSS"][1] < hist["TRAIN_LOSS"][0]
    assert f"_M_{bb.upper()}_H_{H}_" in os.path.basename(res["model_path"])


def test_configuration_outside_the_kernel_envelope_trains_through_the_api(workdir):
    """a pgjanet with 40 hidden units is beyond the HIP kernels: CoreModel builds the ATen restatement (backbones/wide.py) with
    a warning and the same Project flow runs — train_pa of a wide PA, then train_dpd of a kernel-backed DPD (HIP autograd
    bridge) through that ATen PA with torch.optim.AdamW."""
    import warnings
    import opendpd_amd as od
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        kw = dict(dataset_name="DPA_200MHz", PA_backbone="pgjanet", PA_hidden_size=40, frame_length=50, batch_size=256, lr=2e-3, seed=0,
                  accelerator="cuda")
        res = od.train_pa(n_epochs=2, **kw)
        assert any("outside the HIP kernels' envelope" in str(m.message) for m in w)
    assert res["status"] == "completed" and "_M_PGJANET_H_40_" in os.path.basename(res["model_path"])
    hist = pd.read_csv(os.path.join("log", "DPA_200MHz", "train_pa", "history", os.path.basename(res["log_path"])))
    assert len(hist) == 2 and hist["TRAIN_LOSS"][1] < hist["TRAIN_LOSS"][0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = od.train_dpd(n_epochs=1, DPD_backbone="dgru", DPD_hidden_size=8, **kw)
    assert res["status"] == "completed"
    hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
    assert len(hist) == 1 and np.isfinite(hist["TRAIN_LOSS"]).all()


_DP_SCRIPT = """
import os, sys
sys.path.insert(0, {root!r})
os.environ["OPENDPD_DATASETS"] = {datasets!r}
import opendpd_amd as od
kw = dict(dataset_name="DPA_200MHz", PA_backbone="dgru", PA_hidden_size=9, frame_length=50, batch_size=64, lr=1e-3, seed=0,
          accelerator="cuda", n_epochs=2)
res = od.train_pa(**kw)
if {dpd}:
    res = od.train_dpd(DPD_backbone="deltagru_tcnskip", DPD_hidden_size=8, thx=0.01, thh=0.02, **kw)
print("DONE", res["model_path"])
"""


@pytest.mark.parametrize("dpd", [False, True])
def test_data_parallel_api_run_equals_single_process(workdir, dpd):
    """§8e through the API: `torchrun --nproc-per-node 2` over train_pa / train_dpd — every rank draws the same global
    batches, keeps its shard, ONE all-reduce of P+4 floats per step — gives the single-process result (same trajectory up to
    the summation order of the two shard gradients).  Two ranks share this box's one GPU and talk over gloo; on a node with
    one GPU per rank the same code runs over RCCL."""
    import subprocess, sys, shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = _DP_SCRIPT.format(root=root, datasets=os.environ["OPENDPD_DATASETS"], dpd=dpd)
    outs = {}
    for tag, world in (("single", 1), ("dp2", 2)):
        wd = os.path.join(os.getcwd(), f"{tag}_{int(dpd)}")
        os.makedirs(wd)
        open(os.path.join(wd, "run.py"), "w").write(script)
        env = dict(os.environ, OPENDPD_DIST_BACKEND="gloo", OPENDPD_DIST_SINGLE_DEVICE="1")
        cmd = [sys.executable, "run.py"]
        if world > 1:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                   "127.0.0.1", "--master-port", "29531", "run.py"]
        r = subprocess.run(cmd, cwd=wd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "DONE" in r.stdout, r.stderr[-3000:]
        step = "train_dpd" if dpd else "train_pa"
        hist = []
        for dp, _, files in os.walk(os.path.join(wd, "log", "DPA_200MHz", step)):
            hist += [os.path.join(dp, f) for f in files if dp.endswith("history")]
        assert len(hist) == 1, hist                     # rank 0 alone writes the files
        model = [os.path.join(dp, f) for dp, _, files in os.walk(os.path.join(wd, "save", "DPA_200MHz", step)) for f in files]
        assert len(model) == 1
        outs[tag] = (pd.read_csv(hist[0]), torch.load(model[0], map_location="cpu"))
        shutil.rmtree(os.path.join(wd, "log"))
    (h1, m1), (h2, m2) = outs["single"], outs["dp2"]
    assert list(h1.columns) == list(h2.columns) and len(h1) == len(h2) == 2
    assert np.allclose(h1["TRAIN_LOSS"], h2["TRAIN_LOSS"], rtol=2e-3)
    assert np.allclose(h1["VAL_NMSE"], h2["VAL_NMSE"], atol=0.3)       # dB after 2 x 360 steps of diverging rounding
    for k in m1:
        a, b = m1[k].float(), m2[k].float()
        assert (a - b).abs().max() <= 2e-2 * max(1.0, a.abs().max()), k


@pytest.fixture(scope="module")
def apa_workdir(tmp_path_factory):
    wd = tmp_path_factory.mktemp("odpd_apa")
    for tag, name in (("apa200", "APA_200MHz"), ("apa200b", "APA_200MHz_b")):
        d = dict(np.load(os.path.join(GOLDEN, f"{tag}_dataset.npz")))
        ds = wd / "datasets" / name
        ds.mkdir(parents=True)
        (ds / "spec.json").write_text(str(d.pop("spec")))
        for k, v in d.items():
            pd.DataFrame(v, columns=["I", "Q"]).to_csv(ds / f"{k}.csv", index=False)
    old, old_ds = os.getcwd(), os.environ.get("OPENDPD_DATASETS")
    os.chdir(wd)
    os.environ["OPENDPD_DATASETS"] = str(wd / "datasets")
    yield wd
    os.chdir(old)
    if old_ds is not None:
        os.environ["OPENDPD_DATASETS"] = old_ds


@pytest.mark.parametrize("key,ds,bb,cfg", [("dgru_apa200", "APA_200MHz", "dgru", "config2"), ("vdlstm_apa200b", "APA_200MHz_b", "vdlstm", "config4")])
def test_baseline_config_2_and_4_epochs_match_reference_log(apa_workdir, steps_seen, key, ds, bb, cfg):
    """BASELINE configs 2 and 4 on their own datasets: one epoch of train_pa (230 steps of 256 x 200 frames, last batch 157;
    fused single-launch train kernels + native epoch loop) against the row the REFERENCE logged for the same command
    (tests/golden/ref_runs_apa.json, oracle/gen_run_anchors_apa.py), and the reference's own per-step losses of the first 20 steps
    (tests/golden/ref_first_steps.json): float models without thresholds or grids, so all 20 to rounding level."""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_apa.json")))[key]
    res = od.train_pa(dataset_name=ds, PA_backbone=bb, PA_hidden_size=13, frame_length=200, batch_size=256, seed=0, n_epochs=1,
                      accelerator="cuda")
    assert os.path.basename(res["model_path"])[:-3] == ref["model_id"]
    hist = pd.read_csv(os.path.join("log", ds, "train_pa", "history", os.path.basename(res["log_path"])))
    rh = ref["hist"]
    assert list(hist.columns) == list(rh.keys())
    assert hist["N_PARAM"][0] == rh["N_PARAM"][0]
    assert abs(hist["TRAIN_LOSS"][0] - rh["TRAIN_LOSS"][0]) < 2e-3 * rh["TRAIN_LOSS"][0]
    for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_EVM", "TEST_ACLR_AVG"):
        assert abs(hist[col][0] - rh[col][0]) < 0.05, (col, hist[col][0], rh[col][0])   # dB
    _check_first_steps(cfg, steps_seen["losses"], n_exact=20, rel_exact=2e-6, rel_all=2e-6)      # (measured: <= 3.6e-7 on every step)


def test_baseline_config_3_epoch_on_apa_matches_reference_log(apa_workdir, steps_seen):
    """BASELINE config 3 on its own dataset: train_dpd of TRes-DeltaGRU H15 (thx .01, thh .05) in front of the frozen DGRU H23
    PA the REFERENCE trained (its state dict is a fixture), 919 steps of 64 x 200 frames through the cascade step,
    against the row the reference logged (tests/golden/ref_runs_apa.json).  Thresholded model: rounding-level differences
    flip a few delta decisions over 919 steps, hence dB-level tolerances on the metrics."""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_apa.json")))["config3_apa200"]
    m = dict(np.load(os.path.join(GOLDEN, "ref_runs_apa_models.npz")))
    os.makedirs(os.path.dirname(ref["pa_model"]), exist_ok=True)
    torch.save({k[3:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("pa/")}, ref["pa_model"])
    res = od.train_dpd(dataset_name="APA_200MHz", PA_backbone="dgru", PA_hidden_size=23, DPD_backbone="deltagru_tcnskip",
                       DPD_hidden_size=15, thx=0.01, thh=0.05, frame_length=200, batch_size=64, seed=0, n_epochs=1, accelerator="cuda")
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["dpd_model"])
    hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
    rh = ref["hist"]
    assert list(hist.columns) == list(rh.keys())
    assert hist["N_PARAM"][0] == rh["N_PARAM"][0] == 999 + 2751        # the log counts the whole cascade
    print("[config3] gaps:", {c: abs(hist[c][0] - rh[c][0]) / (abs(rh[c][0]) if c == "TRAIN_LOSS" else 1.0)
                              for c in ("TRAIN_LOSS", "SP_T_DX", "SP_T_DH", "VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG")})
    # measured gaps (r04 box): TRAIN_LOSS 6.1e-4 relative, SP_T_DX 0, SP_T_DH 7e-5, metrics <= 0.045 dB — the bounds are ~3 x those (r03: 3 % / 0.5 dB)
    assert abs(hist["TRAIN_LOSS"][0] - rh["TRAIN_LOSS"][0]) < 2e-3 * rh["TRAIN_LOSS"][0]
    assert abs(hist["SP_T_DX"][0] - rh["SP_T_DX"][0]) < 1e-4 and abs(hist["SP_T_DH"][0] - rh["SP_T_DH"][0]) < 3e-4
    for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
        assert abs(hist[col][0] - rh[col][0]) < 0.15, (col, hist[col][0], rh[col][0])
    # the first steps, before any threshold decision has been taken differently: the reference's own per-step losses (measured: <= 2.1e-7 up to
    # step 12, then a first delta decision differs — 192 000 of them per step, each a compare of a rounding-level-different value — and
    # the trajectories part at the 1e-5 .. 5e-4 level)
    _check_first_steps("config3", steps_seen["losses"], n_exact=10)


@pytest.mark.parametrize("key,quant", [("deltajanet", False), ("deltajanet_w8a8", True)])
def test_deltajanet_train_dpd_first_steps_match_the_reference(apa_workdir, steps_seen, key, quant):
    """train_dpd with a deltajanet DPD (float; --quant: INT_Linear fc_out behind the float delta cell) in front of the frozen DGRU H23 PA the
    REFERENCE trained.  The reference trains this epoch and then fails while LOGGING it (modules/paths.py:56 asks the wrapper for
    get_temporal_sparsity, which only its layer defines), so no epoch row exists to compare — its per-step losses of the first 20 steps do
    (oracle/gen_run_anchor_first_steps.py).  Here the epoch completes and the log carries the layer's three sparsity ratios."""
    import opendpd_amd as od
    m = dict(np.load(os.path.join(GOLDEN, "ref_runs_apa_models.npz")))
    pa_rel = json.load(open(os.path.join(GOLDEN, "ref_runs_apa.json")))["config3_apa200"]["pa_model"]
    os.makedirs(os.path.dirname(pa_rel), exist_ok=True)
    torch.save({k[3:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("pa/")}, pa_rel)
    kw = dict(quant=True, n_bits_w=8, n_bits_a=8) if quant else {}
    res = od.train_dpd(dataset_name="APA_200MHz", PA_backbone="dgru", PA_hidden_size=23, DPD_backbone="deltajanet", DPD_hidden_size=12,
                       frame_length=200, batch_size=64, seed=0, n_epochs=1, accelerator="cuda", **kw)
    hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
    assert {"SP_T_DX", "SP_T_DH", "SP_T_DV", "THX", "THH"} <= set(hist.columns)
    assert hist["N_PARAM"][0] == 2751 + 506 + (3 if quant else 0)
    # measured: float <= 4.3e-7, quantised head <= 1.0e-6 on every step (a state next to an activation-grid boundary may round the other way
    # and move a step's loss by ~1e-5: the quantised bound leaves room for that)
    _check_first_steps(key, steps_seen["losses"], n_exact=20, rel_exact=2e-6 if not quant else 5e-5, rel_all=2e-6 if not quant else 5e-5)


# bounds of the W16A16 first-steps check.  Measured (r06, gpurun_out/r06a): step 1 deviates by 2.4e-4, steps 2 .. 20 by 4e-4 .. 7.3e-3 of the
# reference's loss — NOT rounding level from the first step on: 32-bit products summed in fp32 land within rounding reach of a 2^-14 grid
# boundary somewhere among the 64 x 200 x 15 state values of the very first forward pass, the quantised value moves by one grid step and the
# recurrence carries it (tests/test_quant_gpu.py bounds the same effect per output: <= 2.5 LSB on <= 8 %).  The bounds are ~3 x the measured gaps;
# a kernel change that breaks the stage moves these by orders of magnitude (the epoch row's 5 % would not see a 1 % regression).
V2_QAT_EXACT_STEPS, V2_QAT_REL_EXACT, V2_QAT_REL_ALL = 1, 8e-4, 2.2e-2


def test_train_pa_h23_at_a_large_batch_trains_the_same_on_the_split_and_the_exact_kernels(apa_workdir, steps_seen):
    """r06: `train_pa --PA_backbone dgru --PA_hidden_size 23` (the PA of every train_dpd run, bash_scripts/OpenDPDv2.sh:39-52) at a batch that takes the
    16-sequences-per-wave kernels (8 192 frames: 8 steps per epoch on APA_200MHz), three epochs through the API: once on the bf16 x 3 train kernel
    (csrc/gru_s16x.hip, the default), once on the exact-fp32 kernel it replaced (knob "s16x_train" = 0).  Same seed, same frame order: the per-step
    losses agree to 2e-5 (two fp32-equivalent summation orders through 24 optimiser steps), the logged metrics to 0.02 dB, and both runs land on the
    same (still early: 24 optimiser steps) NMSE."""
    import opendpd_amd as od
    from opendpd_amd import _lib
    lib = _lib.load()
    kw = dict(dataset_name="APA_200MHz", PA_backbone="dgru", PA_hidden_size=23, frame_length=200, batch_size=8192, seed=0, n_epochs=3, accelerator="cuda",
              lr=2e-3)
    out = {}
    try:
        for knob in (1, 0):
            assert lib.odpd_set_tuning(b"s16x_train", knob) == 0
            res = od.train_pa(**kw)
            hist = pd.read_csv(os.path.join("log", "APA_200MHz", "train_pa", "history", os.path.basename(res["log_path"])))
            out[knob] = (steps_seen["losses"].cpu().numpy().copy(), hist, torch.load(res["model_path"], map_location="cpu"))
    finally:
        lib.odpd_set_tuning(b"s16x_train", 1)
    # the two runs did take different kernels (their best checkpoints differ in the last bits) and ended in the same place
    pa, pb = out[1][2], out[0][2]
    assert list(pa.keys()) == list(pb.keys())
    assert any(not torch.equal(pa[k], pb[k]) for k in pa)
    for k in pa:
        assert (pa[k] - pb[k]).abs().max() <= 2e-5 * max(1.0, float(pb[k].abs().max())), k
    la, lb = out[1][0], out[0][0]
    assert la.shape == lb.shape and la.shape[0] >= 7
    rel = np.abs(la - lb) / lb
    print("[train_pa H23, batch 8192] last-epoch per-step loss deviation split vs exact:", " ".join(f"{e:.1e}" for e in rel))
    assert rel.max() < 2e-5, rel
    ha, hb = out[1][1], out[0][1]
    assert list(ha.columns) == list(hb.columns) and len(ha) == 3
    for col in ("TRAIN_LOSS",):
        assert np.allclose(ha[col], hb[col], rtol=2e-5), (col, ha[col].values, hb[col].values)
    for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
        assert np.abs(ha[col] - hb[col]).max() < 0.02, (col, ha[col].values, hb[col].values)
    assert ha["TRAIN_LOSS"].iloc[-1] < 0.5 * ha["TRAIN_LOSS"].iloc[0] and ha["VAL_NMSE"].iloc[-1] < -5.0      # (24 steps in all: it learns; measured 0.066 -> 0.0195, -6.6 dB)


def _check_first_steps(key, losses, n_exact, rel_exact=1e-6, rel_all=2e-3):
    """the reference's per-step losses of the first 20 steps (oracle/gen_run_anchor_first_steps.py): the first `n_exact` to rounding level —
    no threshold decision / quantisation boundary has been crossed differently yet —, all 20 within the epoch's tolerance"""
    ref = np.asarray(json.load(open(os.path.join(GOLDEN, "ref_first_steps.json")))[key]["losses"])
    got = losses[:len(ref)].cpu().numpy()
    err = np.abs(got - ref) / ref
    print(f"[{key}] first {len(ref)} steps, relative loss deviation per step:", " ".join(f"{e:.1e}" for e in err))
    assert err[:n_exact].max() <= rel_exact, (key, err)
    assert err.max() <= rel_all, (key, err)


@pytest.fixture
def steps_seen(monkeypatch):
    """per-step losses of the (one) training epoch a test runs: FusedAdamW.last_epoch_losses of the optimiser net_train was given"""
    from opendpd_amd import project
    seen = {}
    inner = project.net_train

    def spy(log, net, loader, optimizer, *a, **k):
        out = inner(log, net, loader, optimizer, *a, **k)
        seen["losses"] = optimizer.last_epoch_losses
        return out
    monkeypatch.setattr(project, "net_train", spy)
    return seen


def test_baseline_config_5_qat_epoch_on_apa_matches_reference_log(apa_workdir, steps_seen):
    """BASELINE config 5 on its own dataset: quantisation-aware train_dpd (QGRU H10, W8A8) in front of the frozen DGRU H23 PA
    the REFERENCE trained, 919 steps of 64 x 200, against the row the reference logged (tests/golden/ref_runs_qat.json,
    oracle/gen_run_anchor_qat.py).  The quantised cell is bit-exact per step (tests/test_quant_gpu.py); the float PA in the loop
    differs at rounding level, which can move a value across a quantisation boundary: dB-level tolerances."""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_qat.json")))
    m = dict(np.load(os.path.join(GOLDEN, "ref_runs_apa_models.npz")))
    os.makedirs(os.path.dirname(ref["pa_model"]), exist_ok=True)
    torch.save({k[3:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("pa/")}, ref["pa_model"])
    res = od.train_dpd(dataset_name="APA_200MHz", PA_backbone="dgru", PA_hidden_size=23, DPD_backbone="qgru", DPD_hidden_size=10,
                       quant=True, n_bits_w=8, n_bits_a=8, frame_length=200, batch_size=64, seed=0, n_epochs=1, accelerator="cuda")
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["dpd_model"])
    hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
    rh = ref["hist"]
    assert list(hist.columns) == list(rh.keys())
    assert hist["N_PARAM"][0] == rh["N_PARAM"][0]
    print("[config5] gaps:", {c: abs(hist[c][0] - rh[c][0]) / (abs(rh[c][0]) if c == "TRAIN_LOSS" else 1.0)
                              for c in ("TRAIN_LOSS", "VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG")})
    # measured gaps (r04 box): TRAIN_LOSS 1.8e-3 relative, metrics <= 0.055 dB (a rounding-level difference in the float PA moves a value across a
    # quantisation boundary now and then) — the bounds are 3 x those
    assert abs(hist["TRAIN_LOSS"][0] - rh["TRAIN_LOSS"][0]) < 6e-3 * rh["TRAIN_LOSS"][0]
    for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
        assert abs(hist[col][0] - rh[col][0]) < 0.2, (col, hist[col][0], rh[col][0])
    _check_first_steps("config5", steps_seen["losses"], n_exact=15)        # (measured: <= 2.9e-7 up to step 19, 3.3e-5 at step 20)
    sd = torch.load(res["model_path"], map_location="cpu")
    want = {k[4:]: v for k, v in np.load(os.path.join(GOLDEN, "ref_runs_qat_models.npz")).items()}
    assert list(sd.keys()) == list(want.keys())


def test_openDPDv2_recipe_on_apa_matches_reference_logs(apa_workdir, steps_seen):
    """bash_scripts/OpenDPDv2.sh:47-117 on APA_200MHz, one epoch per stage, against rows the REFERENCE logged
    (tests/golden/ref_runs_v2.{json,npz}, oracle/gen_run_anchor_opendpdv2.py): float pre-training of TRes-DeltaGRU H15 (thx .01, thh .05,
    lr 5e-3) in front of the frozen DGRU H23 PA the reference trained; then the QAT stage `--quant --n_bits_w 16 --n_bits_a 16
    --quant_dir_label w16a16 --pretrained_model <the REFERENCE's float checkpoint>` (919 steps of 64 x 200 on the quantised delta
    cell, csrc/qat_s16.hip); then run_dpd --quant with the reference's trained quantised weights.  A thresholded model on 16-bit
    grids: rounding-level differences flip delta decisions, hence dB-level tolerances on the training rows; the exported predistorted
    signal (dense deltas at export time, run_dpd.py:60-70) is the quantised network alone."""
    import opendpd_amd as od
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_v2.json")))
    m = dict(np.load(os.path.join(GOLDEN, "ref_runs_v2.npz")))
    pa_path = json.load(open(os.path.join(GOLDEN, "ref_runs_apa.json")))["config3_apa200"]["pa_model"]
    os.makedirs(os.path.dirname(pa_path), exist_ok=True)
    torch.save({k[3:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("pa/")}, pa_path)
    kw = dict(dataset_name="APA_200MHz", PA_backbone="dgru", PA_hidden_size=23, DPD_backbone="deltagru_tcnskip", DPD_hidden_size=15,
              thx=0.01, thh=0.05, frame_length=200, batch_size=64, lr=5e-3, seed=0, n_epochs=1, accelerator="cuda")

    def row(res):
        return pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))

    def compare(hist, rh, n_param):
        assert list(hist.columns) == list(rh.keys())
        assert hist["N_PARAM"][0] == rh["N_PARAM"][0] == n_param
        print("TRAIN_LOSS here / reference:", hist["TRAIN_LOSS"][0], rh["TRAIN_LOSS"][0], {c: (hist[c][0], rh[c][0]) for c in ("SP_T_DX", "SP_T_DH", "VAL_NMSE", "TEST_NMSE")})
        assert abs(hist["TRAIN_LOSS"][0] - rh["TRAIN_LOSS"][0]) < 0.05 * rh["TRAIN_LOSS"][0], (hist["TRAIN_LOSS"][0], rh["TRAIN_LOSS"][0])
        assert abs(hist["SP_T_DX"][0] - rh["SP_T_DX"][0]) < 0.01 and abs(hist["SP_T_DH"][0] - rh["SP_T_DH"][0]) < 0.03
        for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
            assert abs(hist[col][0] - rh[col][0]) < 0.6, (col, hist[col][0], rh[col][0])

    # stage 1: float pre-training
    res = od.train_dpd(**kw)
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["float_stage"]["dpd_model"])
    compare(row(res), ref["float_stage"]["hist"], 999 + 2751)
    # stage 2: QAT from the REFERENCE's float checkpoint (the script picks the newest file matching the float model's ID)
    pre = ref["float_stage"]["dpd_model"]
    torch.save({k[5:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("fdpd/")}, pre)
    q = dict(quant=True, n_bits_w=16, n_bits_a=16, quant_dir_label="w16a16")
    res = od.train_dpd(pretrained_model=pre, **kw, **q)
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["qat_stage"]["dpd_model"])
    assert os.path.normpath(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"]))) == \
        os.path.normpath(ref["qat_stage"]["hist_path"])
    compare(row(res), ref["qat_stage"]["hist"], 1012 + 2751)
    # the QAT stage step by step (r06): the reference's own losses of the first 20 steps of this stage (oracle/gen_run_anchor_first_steps.py,
    # key v2_qat_w16a16).  The epoch row above is chaotic (thresholds on 2^-14 grids: a summation-order change moved TRAIN_LOSS by 10 %), the
    # first steps are not: this is what a change to csrc/qat_s16.hip is judged on
    _check_first_steps("v2_qat_w16a16", steps_seen["losses"], n_exact=V2_QAT_EXACT_STEPS, rel_exact=V2_QAT_REL_EXACT, rel_all=V2_QAT_REL_ALL)
    sd = torch.load(res["model_path"], map_location="cpu")
    ref_sd = {k[5:]: v for k, v in m.items() if k.startswith("qdpd/")}
    assert list(sd.keys()) == list(ref_sd.keys())
    for k, v in ref_sd.items():
        if "_num" in k or "pow2_scale" in k or "n_bits" in k:
            assert np.array_equal(sd[k].numpy(), v), k
    # stage 3: run_dpd --quant with the reference's trained quantised weights
    torch.save({k: torch.from_numpy(v) for k, v in ref_sd.items()}, ref["qat_stage"]["dpd_model"])
    kw_run = {k: v for k, v in kw.items() if k not in ("batch_size", "lr", "n_epochs")}
    out = od.run_dpd(**kw_run, **q)
    assert os.path.normpath(out["output_path"]) == os.path.normpath(ref["run_dpd"]["path"])
    csv = pd.read_csv(out["output_path"])
    assert list(csv.columns) == ref["run_dpd"]["columns"] and len(csv) == ref["run_dpd"]["rows"]
    head = m["run_dpd_head"]
    err = np.abs(csv.to_numpy()[:len(head)] - head).max()
    assert err <= 4 * 2.0 ** -14, err          # 16-bit grids: the fp32 summation order is visible at the level of the output LSB


def test_quantised_gru_dpd_on_dpa_matches_reference_log(workdir):
    """The GRU-swap half of the surgery end to end: train_dpd --DPD_backbone gru --quant --n_bits_w 8 --n_bits_a 8 (H 11, frame 50,
    one epoch) in front of the GRU PA the reference trained, against the reference's logged row and checkpoint layout."""
    import opendpd_amd as od
    os.chdir(workdir)                                             # (the APA tests above moved the process to their own directory)
    os.environ["OPENDPD_DATASETS"] = str(workdir / "datasets")
    ref = json.load(open(os.path.join(GOLDEN, "ref_runs_v2.json")))["gru_w8a8_dpa"]
    m = dict(np.load(os.path.join(GOLDEN, "ref_runs_v2.npz")))
    pa_path = os.path.join("save", "DPA_200MHz", "train_pa", "PA_S_0_M_GRU_H_11_F_50_P_519.pt")
    os.makedirs(os.path.dirname(pa_path), exist_ok=True)
    torch.save({k[7:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("pa_dpa/")}, pa_path)
    res = od.train_dpd(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, DPD_backbone="gru", DPD_hidden_size=11, frame_length=50,
                       batch_size=64, lr=1e-3, seed=0, n_epochs=1, accelerator="cuda", quant=True, n_bits_w=8, n_bits_a=8, quant_dir_label="w8a8")
    assert os.path.normpath(res["model_path"]) == os.path.normpath(ref["dpd_model"])
    hist = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
    rh = ref["hist"]
    assert list(hist.columns) == list(rh.keys()) and hist["N_PARAM"][0] == rh["N_PARAM"][0] == 532 + 519
    assert abs(hist["TRAIN_LOSS"][0] - rh["TRAIN_LOSS"][0]) < 0.02 * rh["TRAIN_LOSS"][0], (hist["TRAIN_LOSS"][0], rh["TRAIN_LOSS"][0])
    for col in ("VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
        assert abs(hist[col][0] - rh[col][0]) < 0.4, (col, hist[col][0], rh[col][0])
    sd = torch.load(res["model_path"], map_location="cpu")
    ref_sd = {k[9:]: v for k, v in m.items() if k.startswith("qgru_dpa/")}
    assert list(sd.keys()) == list(ref_sd.keys())
