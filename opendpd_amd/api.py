"""User-facing API with the reference's `opendpd.api` signatures and return dictionaries (opendpd/api.py:27-503):
train_pa, train_dpd, run_dpd, load_dataset, create_dataset, OpenDPDTrainer."""
import json
import os

import pandas as pd

from . import data as D
from .project import Project, run_run_dpd, run_train_dpd, run_train_pa


def _need_name(fn, dataset_name, dataset_path):
    if dataset_path:
        raise ValueError(f"{fn} no longer accepts dataset_path. Please create an OpenDPD dataset (e.g., with "
                         f"create_dataset) and pass its dataset_name instead.")
    if not dataset_name:
        raise ValueError(f"{fn} requires dataset_name. Create a dataset first with create_dataset().")


def train_pa(dataset_name=None, dataset_path=None, PA_backbone="gru", PA_hidden_size=23, n_epochs=100, batch_size=256,
             lr=5e-4, accelerator="cpu", frame_length=200, seed=0, **kwargs):
    _need_name("train_pa", dataset_name, dataset_path)
    proj = Project(step="train_pa", dataset_name=dataset_name, PA_backbone=PA_backbone, PA_hidden_size=PA_hidden_size,
                   n_epochs=n_epochs, batch_size=batch_size, lr=lr, accelerator=accelerator, frame_length=frame_length,
                   seed=seed, **kwargs)
    run_train_pa(proj)
    return {"status": "completed", "model_path": proj.path_save_file_best, "log_path": proj.path_log_file_best}


def train_dpd(dataset_name=None, dataset_path=None, DPD_backbone="deltagru_tcnskip", DPD_hidden_size=15, PA_backbone="gru",
              PA_hidden_size=23, n_epochs=100, batch_size=256, lr=5e-4, accelerator="cpu", frame_length=200, seed=0,
              thx=0.0, thh=0.0, **kwargs):
    _need_name("train_dpd", dataset_name, dataset_path)
    proj = Project(step="train_dpd", dataset_name=dataset_name, DPD_backbone=DPD_backbone, DPD_hidden_size=DPD_hidden_size,
                   PA_backbone=PA_backbone, PA_hidden_size=PA_hidden_size, n_epochs=n_epochs, batch_size=batch_size, lr=lr,
                   accelerator=accelerator, frame_length=frame_length, seed=seed, thx=thx, thh=thh, **kwargs)
    run_train_dpd(proj)
    return {"status": "completed", "model_path": proj.path_save_file_best, "log_path": proj.path_log_file_best}


def run_dpd(dataset_name=None, dataset_path=None, DPD_backbone="deltagru_tcnskip", DPD_hidden_size=15, accelerator="cpu", **kwargs):
    _need_name("run_dpd", dataset_name, dataset_path)
    proj = Project(step="run_dpd", dataset_name=dataset_name, DPD_backbone=DPD_backbone, DPD_hidden_size=DPD_hidden_size,
                   accelerator=accelerator, **kwargs)
    return {"status": "completed", "output_path": run_run_dpd(proj)}


def load_dataset(dataset_path):
    """-> dict of the six arrays (opendpd/api.py:263-313)."""
    keys = ("X_train", "y_train", "X_val", "y_val", "X_test", "y_test")
    return dict(zip(keys, D.load_dataset(dataset_path=dataset_path)))


def create_dataset(csv_path, output_dir, dataset_name, train_ratio=0.6, val_ratio=0.2, test_ratio=0.2, dataset_format="single_csv",
                   csv_filename=None, **spec_kwargs):
    """Import a CSV with columns I_in,Q_in,I_out,Q_out as an OpenDPD dataset directory with a spec.json — the same files, byte for
    byte, as opendpd/api.py:316-431 writes (spec key order and defaults `nperseg` 2560 / `n_sub_ch` 1, `split_indices` of the
    single-CSV layout, column names of the six-file layout, no check that the ratios add up).  Returns the resolved directory."""
    fmt = dataset_format.lower()
    if fmt not in ("single_csv", "split_csv"):
        raise ValueError("dataset_format must be 'single_csv' or 'split_csv'")
    out = os.path.realpath(os.path.join(os.path.expanduser(str(output_dir)), dataset_name))
    os.makedirs(out, exist_ok=True)
    df = pd.read_csv(os.path.expanduser(str(csv_path)))
    need = ["I_in", "Q_in", "I_out", "Q_out"]
    if not all(c in df.columns for c in need):
        raise ValueError(f"CSV must contain columns: {need}. Found: {df.columns.tolist()}")
    n_train, n_val = int(len(df) * train_ratio), int(len(df) * val_ratio)
    parts = {"train": df.iloc[:n_train], "val": df.iloc[n_train:n_train + n_val], "test": df.iloc[n_train + n_val:]}
    spec = {"split_ratios": {"train": train_ratio, "val": val_ratio, "test": test_ratio}, "nperseg": 2560, "n_sub_ch": 1}
    if fmt == "split_csv":
        for split, part in parts.items():
            part[["I_in", "Q_in"]].to_csv(os.path.join(out, f"{split}_input.csv"), index=False)
            part[["I_out", "Q_out"]].to_csv(os.path.join(out, f"{split}_output.csv"), index=False)
    else:
        csv_filename = csv_filename or "data.csv"
        pd.concat(list(parts.values()), axis=0).to_csv(os.path.join(out, csv_filename), index=False)
        spec["csv_filename"] = csv_filename
        spec["split_indices"] = {"train_end": len(parts["train"]), "val_end": len(parts["train"]) + len(parts["val"])}
    spec.update(spec_kwargs)
    spec["dataset_format"] = fmt
    with open(os.path.join(out, "spec.json"), "w") as f:
        json.dump(spec, f, indent=4)
    print(f"Dataset created successfully at: {out}")
    for label, split in (("Training", "train"), ("Validation", "val"), ("Test", "test")):
        print(f"  - {label} samples: {len(parts[split])}")
    return out


class OpenDPDTrainer:
    """Object-style front end (opendpd/api.py:434-503): a stored configuration merged under per-call keyword arguments;
    `train_dpd` trains the PA first when that has not happened yet, `run` refuses before `train_dpd`."""

    def __init__(self, dataset_name=None, dataset_path=None, **kwargs):
        self.dataset_name = dataset_name
        self.dataset_path = dataset_path
        self.config = kwargs
        self.pa_trained = False
        self.dpd_trained = False

    def _config(self, kwargs):
        config = {**self.config, **kwargs}
        if self.dataset_name:
            config["dataset_name"] = self.dataset_name
        elif self.dataset_path:
            config["dataset_path"] = self.dataset_path
        return config

    def train_pa(self, **kwargs):
        result = train_pa(**self._config(kwargs))
        self.pa_trained = True
        return result

    def train_dpd(self, **kwargs):
        if not self.pa_trained:
            print("Warning: PA model not trained yet. Training PA model first...")
            self.train_pa()
        result = train_dpd(**self._config(kwargs))
        self.dpd_trained = True
        return result

    def run(self, **kwargs):
        if not self.dpd_trained:
            raise RuntimeError("DPD model not trained yet. Call train_dpd() first.")
        return run_dpd(**self._config(kwargs))
