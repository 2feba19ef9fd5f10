"""CPU checks of the opendpd.api mirror: argument contract, dataset creation/loading, shuffle order of the on-device
frame loader (identical to a reference DataLoader(shuffle=True) over materialised frames)."""
import os

import numpy as np
import pandas as pd
import pytest
import torch
from torch.utils.data import DataLoader


def test_api_argument_contract():
    import opendpd_amd as od
    with pytest.raises(ValueError):
        od.train_pa(dataset_path="/tmp/x.csv")
    with pytest.raises(ValueError):
        od.train_pa()
    with pytest.raises(ValueError):
        od.train_dpd(dataset_path="/tmp/x.csv")
    with pytest.raises(ValueError):
        od.run_dpd()


def test_create_and_load_dataset_write_the_reference_files(tmp_path, capsys):
    """opendpd.api.create_dataset / load_dataset (api.py:263-431): every file byte for byte what the REFERENCE's function wrote for
    the same CSV and arguments (tests/golden/create_dataset_ref.json, oracle/gen_golden_api.py) — spec.json key order, defaults and
    split_indices, column names of the six-file layout, case-insensitive format name — and the same six arrays loaded back."""
    import json
    import os
    import opendpd_amd as od
    from tests.golden_util import GOLDEN
    ref = json.load(open(os.path.join(GOLDEN, "create_dataset_ref.json")))
    csv = tmp_path / "in.csv"
    csv.write_text(ref["csv"])
    for name, case in ref["cases"].items():
        d = od.create_dataset(str(csv), str(tmp_path / name), "MyPA", **case["kwargs"])
        assert d == os.path.realpath(str(tmp_path / name / "MyPA"))
        assert sorted(os.listdir(d)) == sorted(case["files"])
        for f, text in case["files"].items():
            assert open(os.path.join(d, f)).read() == text, (name, f)
        loaded = od.load_dataset(d)
        assert list(loaded) == list(case["loaded"])
        for k, v in case["loaded"].items():
            assert np.array_equal(np.asarray(loaded[k]), np.asarray(v)), (name, k)
    assert "Dataset created successfully at:" in capsys.readouterr().out
    with pytest.raises(ValueError):
        od.create_dataset(str(csv), str(tmp_path / "x"), "MyPA", dataset_format="parquet")
    pd.DataFrame(np.zeros((4, 2)), columns=["I", "Q"]).to_csv(tmp_path / "bad.csv", index=False)
    with pytest.raises(ValueError):
        od.create_dataset(str(tmp_path / "bad.csv"), str(tmp_path / "y"), "MyPA")


def test_device_frame_loader_matches_reference_dataloader_order():
    from opendpd_amd.data import IQFrameDataset
    from opendpd_amd.project import DeviceFrameLoader
    rng = np.random.RandomState(1)
    x, y = rng.randn(500, 2), rng.randn(500, 2)
    torch.manual_seed(0)
    ref = list(DataLoader(IQFrameDataset(x, y, 50, 1), batch_size=64, shuffle=True))
    torch.manual_seed(0)
    ours = list(DeviceFrameLoader(x, y, 50, 1, 64, torch.device("cpu"), shuffle=True))
    assert len(ref) == len(ours)
    for (a, b), (c, d) in zip(ref, ours):
        assert torch.equal(a, c) and torch.equal(b, d)
    # the RNG state after one epoch is the same too (the next epoch's permutation will match)
    torch.manual_seed(0)
    list(DataLoader(IQFrameDataset(x, y, 50, 1), batch_size=64, shuffle=True)); r1 = torch.rand(1)
    torch.manual_seed(0)
    list(DeviceFrameLoader(x, y, 50, 1, 64, torch.device("cpu"), shuffle=True)); r2 = torch.rand(1)
    assert torch.equal(r1, r2)
    # three consecutive epochs, and the unshuffled loader (whose iterator still draws its base seed)
    for shuffle in (True, False):
        torch.manual_seed(5)
        dl = DataLoader(IQFrameDataset(x, y, 50, 1), batch_size=64, shuffle=shuffle)
        ref = [[a.clone() for a, _ in dl] for _ in range(3)]
        r1 = torch.rand(1)
        torch.manual_seed(5)
        mine = DeviceFrameLoader(x, y, 50, 1, 64, torch.device("cpu"), shuffle=shuffle)
        ours = [[a.clone() for a, _ in mine] for _ in range(3)]
        r2 = torch.rand(1)
        assert torch.equal(r1, r2)
        for ea, eb in zip(ref, ours):
            assert len(ea) == len(eb) and all(torch.equal(a, b) for a, b in zip(ea, eb))


def test_without_a_hip_device_every_accelerator_is_refused():
    """There is no CPU path: on a box without a HIP device (this container) the reference's default accelerator='cpu' and 'cuda' both
    raise; on a GPU box 'cpu' is mapped to the HIP device with a warning (tests/test_e2e_gpu.py)."""
    import pytest
    import torch
    from opendpd_amd.project import Project
    if torch.cuda.is_available():
        pytest.skip("needs a box without a HIP device")
    for acc in ("cpu", "cuda"):
        proj = Project.__new__(Project)
        proj.accelerator, proj.devices = acc, 0
        with pytest.raises(ValueError, match="no CPU fallback"):
            proj.set_device()
    proj.accelerator = "mps"
    with pytest.raises(ValueError, match="not supported"):
        proj.set_device()


def test_lr_scheduler_mirrors_torch_reduce_on_plateau():
    """project.py:289-296 steps ReduceLROnPlateau(min, factor, patience, threshold 1e-4, min_lr) on NMSE / ACLR in dB — negative
    numbers, where torch's relative threshold admits slightly worse values as improvements.  Same LR trajectory as torch's class
    on dB-like, positive and mixed sequences."""
    import numpy as np
    import torch
    from opendpd_amd.project import ReduceLROnPlateau

    class Opt:
        def __init__(self, lr):
            self.param_groups = [{"lr": lr}]

    rng = np.random.RandomState(0)
    seqs = [(-20 - 3 * np.abs(np.sin(np.arange(60) / 3.0)) + 0.002 * rng.randn(60)).tolist(),      # dB plateau with tiny wobble
            (-25 + 0.0015 * np.arange(40)).tolist(),                                                  # each value < 0.01 % worse
            np.abs(rng.randn(50)).tolist(), (rng.randn(50) * 0.5).tolist(), [1.0] * 30, [-30.0] * 30]
    for factor, patience, min_lr in ((0.1, 10, 1e-4), (0.5, 2, 1e-5), (0.3, 0, 1e-3)):
        for seq in seqs:
            w = torch.nn.Parameter(torch.zeros(1))
            topt = torch.optim.SGD([w], lr=5e-3)
            tsch = torch.optim.lr_scheduler.ReduceLROnPlateau(topt, mode="min", factor=factor, patience=patience, threshold=1e-4,
                                                              min_lr=min_lr)
            mine = Opt(5e-3)
            msch = ReduceLROnPlateau(mine, factor, patience, min_lr)
            for v in seq:
                tsch.step(v)
                msch.step(v)
                assert mine.param_groups[0]["lr"] == topt.param_groups[0]["lr"], (factor, patience, v)


def test_optimizer_choices_follow_the_reference():
    """project.py:274-287: adamw / adam / sgd(momentum 0.9) / rmsprop with torch's defaults — the fused HIP optimiser of that kind for a
    kernel-backed model, torch's own class for a configuration beyond the kernels' envelope; adabound imports a package the reference
    does not ship; anything else raises like the reference."""
    import warnings
    import pytest
    import torch
    from types import SimpleNamespace
    from opendpd_amd import CoreModel
    from opendpd_amd.project import Project
    from opendpd_amd.train_funcs import FusedAdam, FusedAdamW, FusedRMSprop, FusedSGD
    net = CoreModel(2, 8, 1, "gru")
    mk = lambda t: SimpleNamespace(opt_type=t, lr=2e-3, decay_factor=0.1, patience=10, lr_end=1e-4)
    opt, sch = Project.build_optimizer(mk("sgd"), net)
    assert isinstance(opt, FusedSGD) and opt.kind == "sgd" and opt.param_groups[0]["momentum"] == 0.9 and opt.param_groups[0]["lr"] == 2e-3
    assert isinstance(Project.build_optimizer(mk("adam"), net)[0], FusedAdam)
    assert isinstance(Project.build_optimizer(mk("rmsprop"), net)[0], FusedRMSprop)
    opt, sch = Project.build_optimizer(mk("adamw"), net)
    assert isinstance(opt, FusedAdamW) and opt.kind == "adamw" and opt.param_groups[0]["lr"] == 2e-3 and opt.param_groups[0]["weight_decay"] == 0.01
    assert (sch.factor, sch.patience, sch.min_lr) == (0.1, 10, 1e-4)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wide = CoreModel(2, 40, 1, "pgjanet")       # beyond the envelope: ATen restatement, torch's optimisers
    opt, _ = Project.build_optimizer(mk("sgd"), wide)
    assert isinstance(opt, torch.optim.SGD) and opt.param_groups[0]["momentum"] == 0.9 and opt.param_groups[0]["lr"] == 2e-3
    assert isinstance(Project.build_optimizer(mk("adam"), wide)[0], torch.optim.Adam)
    assert isinstance(Project.build_optimizer(mk("rmsprop"), wide)[0], torch.optim.RMSprop)
    assert isinstance(Project.build_optimizer(mk("adamw"), wide)[0], torch.optim.AdamW)
    with pytest.raises(ModuleNotFoundError):
        Project.build_optimizer(mk("adabound"), net)
    with pytest.raises(RuntimeError):
        Project.build_optimizer(mk("lion"), net)


def test_project_defaults_equal_the_reference_cli_defaults():
    """arguments.py:8-89 (values and types), captured by oracle/gen_golden_api.py"""
    import json
    import os
    from opendpd_amd.project import DEFAULTS
    from tests.golden_util import GOLDEN
    ref = json.load(open(os.path.join(GOLDEN, "argument_defaults.json")))
    assert sorted(ref) == sorted(DEFAULTS)
    for k, v in ref.items():
        assert DEFAULTS[k] == v and type(DEFAULTS[k]) is type(v), (k, DEFAULTS[k], v)


def test_public_api_signatures_equal_the_reference():
    """opendpd/api.py: same parameter names, order, defaults and kinds (a positional call binds the same way), same public
    methods on OpenDPDTrainer — captured from the reference by oracle/gen_golden_api.py"""
    import inspect
    import json
    import os
    import opendpd_amd.api as api
    from tests.golden_util import GOLDEN
    ref = json.load(open(os.path.join(GOLDEN, "api_signatures.json")))
    sig = lambda f: [[n, None if q.default is inspect._empty else q.default, str(q.kind)] for n, q in inspect.signature(f).parameters.items()]
    for name, want in ref.items():
        if name == "OpenDPDTrainer.public":
            assert [n for n in dir(api.OpenDPDTrainer) if not n.startswith("_")] == want
            continue
        obj = api
        for part in name.split("."):
            obj = getattr(obj, part)
        assert sig(obj) == want, name


def test_trainer_object_follows_the_reference_flow(monkeypatch, capsys):
    """api.py:449-503: stored config under per-call kwargs, dataset_name wins over dataset_path, train_dpd trains the PA first,
    run() refuses before train_dpd"""
    import opendpd_amd.api as api
    calls = []
    monkeypatch.setattr(api, "train_pa", lambda **kw: calls.append(("pa", kw)) or {"status": "completed"})
    monkeypatch.setattr(api, "train_dpd", lambda **kw: calls.append(("dpd", kw)) or {"status": "completed"})
    monkeypatch.setattr(api, "run_dpd", lambda **kw: calls.append(("run", kw)) or {"status": "completed"})
    t = api.OpenDPDTrainer(dataset_name="DS", dataset_path="/x", n_epochs=3, lr=1e-3)
    with pytest.raises(RuntimeError):
        t.run()
    t.train_dpd(lr=2e-3, DPD_backbone="gmp")
    assert "PA model not trained yet" in capsys.readouterr().out
    assert calls[0] == ("pa", {"n_epochs": 3, "lr": 1e-3, "dataset_name": "DS"})
    assert calls[1] == ("dpd", {"n_epochs": 3, "lr": 2e-3, "DPD_backbone": "gmp", "dataset_name": "DS"})
    t.run(accelerator="cuda")
    assert calls[2] == ("run", {"n_epochs": 3, "lr": 1e-3, "accelerator": "cuda", "dataset_name": "DS"})
    t2 = api.OpenDPDTrainer(dataset_path="/x")
    t2.train_pa()
    assert calls[3] == ("pa", {"dataset_path": "/x"}) and t2.pa_trained and not t2.dpd_trained


def test_csv_logger_appends_the_bytes_pandas_would_rewrite(tmp_path):
    """modules/loggers.py:119-163 rewrites the history CSV through pandas after every epoch; CsvLogger appends the new line instead (and writes
    the best-row file directly): after every epoch both files equal the pandas rewrite byte for byte — quoting, nan, integer columns, negative
    zero included; a row holding something the fast writer does not take (a float32 scalar) goes through pandas itself"""
    import numpy as np
    import pandas as pd
    from opendpd_amd.project import CsvLogger
    lg = CsvLogger(str(tmp_path / "m.pt"), str(tmp_path / "best.csv"), str(tmp_path / "hist.csv"))
    rows = []
    for ep in range(6):
        stat = {"EPOCH": ep, "N_EPOCH": 6, "TIME:": 0.0123 * ep, "LR": 5e-4, "BATCH_SIZE": 64, "N_PARAM": np.int64(1041), "BACKBONE": 'dgru, "x"',
                "HIDDEN_SIZE": 13, "TRAIN_LOSS": float(np.float64(1e-3 / (ep + 1))), "VAL_NMSE": float("nan") if ep == 2 else -30.123456789 * ep,
                "VAL_EVM": -1e-12}
        if ep == 4:
            stat["VAL_EVM"] = np.float32(0.25)         # not a Python float: the reference's formatter leaves it to pandas
        lg.write_log(stat)
        lg._write_best(ep)
        rows.append(["{:.8f}".format(v) if isinstance(v, float) else v for v in stat.values()])
        pd.DataFrame(rows, columns=list(stat.keys())).to_csv(tmp_path / "ref.csv", index=False)
        pd.DataFrame([rows[ep]], columns=list(stat.keys())).to_csv(tmp_path / "refb.csv", index=False)
        assert (tmp_path / "ref.csv").read_bytes() == (tmp_path / "hist.csv").read_bytes(), ep
        assert (tmp_path / "refb.csv").read_bytes() == (tmp_path / "best.csv").read_bytes(), ep
