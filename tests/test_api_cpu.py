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


def test_create_and_load_dataset(tmp_path):
    import opendpd_amd as od
    rng = np.random.RandomState(0)
    df = pd.DataFrame(rng.randn(100, 4), columns=["I_in", "Q_in", "I_out", "Q_out"])
    csv = tmp_path / "mypa.csv"
    df.to_csv(csv, index=False)
    for fmt in ("single_csv", "split_csv"):
        p = od.create_dataset(str(csv), output_dir=str(tmp_path / fmt), dataset_name="MyPA", dataset_format=fmt,
                              input_signal_fs=800e6, bw_main_ch=200e6, n_sub_ch=10, nperseg=16)
        d = od.load_dataset(p)
        assert d["X_train"].shape == (60, 2) and d["X_val"].shape == (20, 2) and d["y_test"].shape == (20, 2)
        assert np.allclose(d["X_train"], df[["I_in", "Q_in"]].to_numpy()[:60])
    with pytest.raises(ValueError):
        od.create_dataset(str(csv), train_ratio=0.5, val_ratio=0.2, test_ratio=0.2)


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


def test_cpu_accelerator_is_refused():
    from opendpd_amd.project import Project
    os.environ.pop("OPENDPD_DATASETS", None)
