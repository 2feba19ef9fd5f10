"""Lockstep sweeps (opendpd_amd/sweep.py, csrc odpd_train_epoch_sweep / odpd_backbone_fwd_sweep): K train_pa runs of the reference's seed sweep
(bash_scripts/train_all_pa.sh:26-57) trained together must leave exactly the files K solo runs leave — every logged column but the wall
clock identical, the saved state dicts bit-identical — and must actually take the one-launch-per-step path."""
import ctypes as C
import os

import numpy as np
import pandas as pd
import pytest
import torch

from tests.golden_util import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture()
def workdir(tmp_path):
    d = dict(np.load(os.path.join(GOLDEN, "dpa200_dataset.npz")))
    ds = tmp_path / "datasets" / "DPA_200MHz"
    ds.mkdir(parents=True)
    (ds / "spec.json").write_text(str(d.pop("spec")))
    for k, v in d.items():
        pd.DataFrame(v, columns=["I", "Q"]).to_csv(ds / f"{k}.csv", index=False)
    old = os.getcwd()
    os.chdir(tmp_path)
    os.environ["OPENDPD_DATASETS"] = str(tmp_path / "datasets")
    yield tmp_path
    os.chdir(old)


def _same_files(solo, swept):
    hs, hw = pd.read_csv(solo["log_path"].replace("best", "history")), pd.read_csv(swept["log_path"].replace("best", "history"))
    assert list(hs.columns) == list(hw.columns) and len(hs) == len(hw)
    for col in hs.columns:
        if col != "TIME:":
            assert hs[col].equals(hw[col]), (col, hs[col].tolist(), hw[col].tolist())
    sa, sb = torch.load(solo["model_path"], map_location="cpu"), torch.load(swept["model_path"], map_location="cpu")
    assert list(sa) == list(sb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize("bb,H,batch,T", [("dgru", 13, 64, 50), ("gru", 23, 256, 200)])
def test_swept_runs_equal_their_solo_runs_bit_for_bit(workdir, bb, H, batch, T):
    import opendpd_amd as od
    seeds = (0, 1, 2)
    kw = dict(dataset_name="DPA_200MHz", PA_backbone=bb, PA_hidden_size=H, n_epochs=3, batch_size=batch, frame_length=T, lr=1e-3,
              accelerator="cuda")
    os.makedirs("solo", exist_ok=True)
    os.chdir("solo")
    solo = [od.train_pa(seed=s, **kw) for s in seeds]
    solo = [{k: (os.path.abspath(v) if k.endswith("_path") else v) for k, v in r.items()} for r in solo]
    os.chdir("..")
    os.makedirs("swept", exist_ok=True)
    os.chdir("swept")
    swept = od.train_pa_sweep(seeds=seeds, **kw)
    assert len(swept) == len(seeds) and all(r["lockstep"] for r in swept)      # the one-launch-per-step path carried them
    for a, b in zip(solo, swept):
        _same_files(a, b)


def test_mixed_hidden_sizes_form_one_group_per_shape(workdir):
    import opendpd_amd as od
    kw = dict(dataset_name="DPA_200MHz", PA_backbone="dgru", n_epochs=2, batch_size=64, frame_length=50, lr=1e-3, accelerator="cuda")
    swept = od.train_pa_sweep(seeds=(0, 1), hidden_sizes=(8, 13), **kw)
    assert [(r["PA_hidden_size"], r["seed"]) for r in swept] == [(8, 0), (8, 1), (13, 0), (13, 1)]
    os.makedirs("solo", exist_ok=True)
    os.chdir("solo")
    solo = od.train_pa(seed=1, PA_hidden_size=13, **kw)
    solo = {k: (os.path.abspath(v) if k.endswith("_path") else v) for k, v in solo.items()}
    os.chdir("..")
    _same_files(solo, swept[3])


def test_a_family_without_sweep_kernels_still_runs_in_the_lockstep_loop(workdir):
    """vdlstm (train_all_pa.sh's second backbone) has no sweep launch: its runs advance through the ordinary per-run epoch inside the same
    loop, and still equal their solo runs (the per-run RNG copies keep their shuffles apart)"""
    import opendpd_amd as od
    kw = dict(dataset_name="DPA_200MHz", PA_backbone="vdlstm", PA_hidden_size=8, n_epochs=2, batch_size=64, frame_length=50, lr=1e-3,
              accelerator="cuda")
    swept = od.train_pa_sweep(seeds=(0, 1), **kw)
    assert not any(r["lockstep"] for r in swept)
    os.makedirs("solo", exist_ok=True)
    os.chdir("solo")
    solo = od.train_pa(seed=1, **kw)
    solo = {k: (os.path.abspath(v) if k.endswith("_path") else v) for k, v in solo.items()}
    os.chdir("..")
    _same_files(solo, swept[1])


def test_entry_points_refuse_what_they_do_not_serve():
    from opendpd_amd import _lib
    lib = _lib.load()
    d = _lib.ModelDesc(_lib.BACKBONE_IDS["dgru"], 13, 0.0, 0.0, 0, 0, 0)
    assert lib.odpd_sweep_train_supported(C.byref(d), 256, 200) == 1 and lib.odpd_sweep_fwd_supported(C.byref(d), 2, 19662) == 1
    assert lib.odpd_sweep_train_supported(C.byref(d), 65536, 200) == 0                      # the 16-sequences-per-wave regime fills the chip alone
    v = _lib.ModelDesc(_lib.BACKBONE_IDS["vdlstm"], 13, 0.0, 0.0, 0, 0, 0)
    assert lib.odpd_sweep_train_supported(C.byref(v), 256, 200) == 0
    assert lib.odpd_sweep_scratch_bytes(0, 10) < 0


def test_throughput_mode_equals_the_solo_run_with_the_same_kernel_forced(workdir):
    """exact=False: the training steps on the 16-sequences-per-wave kernel.  Bit-identical to the solo run with that kernel forced
    (odpd_set_tuning("s16_min_batch", 0)), and within float tolerance of the default solo run (another summation order: first-epoch TRAIN_LOSS
    to 1e-5, NMSE to 0.05 dB)."""
    import opendpd_amd as od
    from opendpd_amd import _lib
    lib = _lib.load()
    seeds = (0, 1, 2, 3)
    kw = dict(dataset_name="DPA_200MHz", PA_backbone="dgru", PA_hidden_size=13, n_epochs=2, batch_size=64, frame_length=50, lr=1e-3, accelerator="cuda")
    os.makedirs("swept", exist_ok=True)
    os.chdir("swept")
    swept = od.train_pa_sweep(seeds=seeds, exact=False, **kw)
    assert all(r["lockstep"] and r["mode"] == "s16" for r in swept)
    swept = [{k: (os.path.abspath(v) if k.endswith("_path") else v) for k, v in r.items()} for r in swept]
    os.chdir("..")
    os.makedirs("forced", exist_ok=True)
    os.chdir("forced")
    try:
        assert lib.odpd_set_tuning(b"s16_min_batch", 0) == 0
        forced = od.train_pa(seed=2, **kw)
    finally:
        lib.odpd_set_tuning(b"s16_min_batch", -1)
    forced = {k: (os.path.abspath(v) if k.endswith("_path") else v) for k, v in forced.items()}
    os.chdir("..")
    hs, hw = pd.read_csv(forced["log_path"].replace("best", "history")), pd.read_csv(swept[2]["log_path"].replace("best", "history"))
    assert hs["TRAIN_LOSS"].equals(hw["TRAIN_LOSS"])           # the training trajectory is the forced solo run's, bit for bit
    os.makedirs("solo", exist_ok=True)
    os.chdir("solo")
    solo = od.train_pa(seed=2, **kw)
    hd = pd.read_csv(solo["log_path"].replace("best", "history"))
    assert abs(hd["TRAIN_LOSS"][0] - hw["TRAIN_LOSS"][0]) < 1e-5 * hd["TRAIN_LOSS"][0] + 1e-8
    assert abs(hd["VAL_NMSE"][1] - hw["VAL_NMSE"][1]) < 0.05
