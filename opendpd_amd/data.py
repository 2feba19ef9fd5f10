"""Dataset loading and framing: restates modules/data_collector.py (load_dataset :17-140, IQSegmentDataset :203-230,
IQFrameDataset :233-252) and utils/util.py:18-33 (target gain).

Datasets are directories with train/val/test `{input,output}.csv` (columns I,Q) + spec.json, or one CSV with columns
I_in,Q_in,I_out,Q_out (+ split ratios).  `dataset_name` is looked up under $OPENDPD_DATASETS, ./datasets, then the
package's own datasets/ directory.  Frames are materialised exactly like the reference ((N-F)//s+1 windows, fp32)."""
import json
import os
from pathlib import Path

import numpy as np
import pandas as pd
import torch
from torch.utils.data import Dataset

_PKG_DIR = Path(__file__).resolve().parent.parent


def dataset_roots():
    roots = []
    if os.environ.get("OPENDPD_DATASETS"):
        roots.append(Path(os.environ["OPENDPD_DATASETS"]))
    roots += [Path.cwd() / "datasets", _PKG_DIR / "datasets"]
    return roots


def resolve_dataset(dataset_name=None, dataset_path=None):
    if dataset_name:
        for r in dataset_roots():
            if (r / dataset_name).exists():
                return r / dataset_name
        raise FileNotFoundError(f"dataset '{dataset_name}' not found under {[str(r) for r in dataset_roots()]}")
    if dataset_path:
        p = Path(dataset_path).expanduser()
        return p if p.is_absolute() else (Path.cwd() / p).resolve()
    raise ValueError("Either dataset_name or dataset_path must be provided")


def _split_frame(df, train_ratio, val_ratio):
    need = ["I_in", "Q_in", "I_out", "Q_out"]
    if not all(c in df.columns for c in need):
        raise ValueError(f"CSV must contain columns: {need}. Found: {df.columns.tolist()}")
    n = len(df)
    a, b = int(n * train_ratio), int(n * train_ratio) + int(n * val_ratio)
    parts = (df.iloc[:a], df.iloc[a:b], df.iloc[b:])
    out = []
    for p in parts:
        out += [p[["I_in", "Q_in"]].to_numpy(), p[["I_out", "Q_out"]].to_numpy()]
    return tuple(out)


def load_spec(dataset_name=None, dataset_path=None):
    p = resolve_dataset(dataset_name, dataset_path)
    if p.is_file():
        return {"dataset_format": "single_csv", "split_ratios": {"train": 0.6, "val": 0.2, "test": 0.2}, "nperseg": 2560}
    sp = p / "spec.json"
    if not sp.exists():
        raise FileNotFoundError(f"spec.json not found in dataset path: {p}")
    return json.load(open(sp))


_share = None      # sweeps (opendpd_amd/sweep.py): {resolved path: arrays} while K runs of one dataset are set up — one parse of the CSVs, not K


def load_dataset(dataset_name=None, dataset_path=None):
    """-> (X_train, y_train, X_val, y_val, X_test, y_test), float64 (N,2) arrays."""
    p = resolve_dataset(dataset_name, dataset_path)
    if _share is not None:
        key = str(p)
        if key not in _share:
            _share[key] = _load_dataset(p)
        return tuple(a.copy() for a in _share[key])
    return _load_dataset(p)


def _load_dataset(p):
    if p.is_file() and p.suffix.lower() == ".csv":
        return _split_frame(pd.read_csv(p), 0.6, 0.2)
    spec = json.load(open(p / "spec.json")) if (p / "spec.json").exists() else {}
    if spec.get("dataset_format", "split_csv") == "single_csv":
        r = spec.get("split_ratios", {"train": 0.6, "val": 0.2, "test": 0.2})
        return _split_frame(pd.read_csv(p / spec.get("csv_filename", "data.csv")), r.get("train", 0.6), r.get("val", 0.2))
    return tuple(pd.read_csv(p / f"{s}_{k}.csv").to_numpy() for s in ("train", "val", "test") for k in ("input", "output"))


def set_target_gain(x, y):
    amp = lambda v: np.sqrt(v[:, 0] ** 2 + v[:, 1] ** 2)
    return np.mean(np.max(amp(y)) / np.max(amp(x)))      # a numpy scalar of the data's dtype, as utils/util.py:26-33 returns


def frames(sequence, frame_length, stride=1):
    """All length-F windows at the given stride as an fp32 tensor (n, F, 2)."""
    seq = np.asarray(sequence)
    n = (len(seq) - frame_length) // stride + 1
    idx = (np.arange(n) * stride)[:, None] + np.arange(frame_length)[None, :]
    return torch.from_numpy(seq[idx].astype(np.float32))


def segments(sequence, nperseg):
    """Non-overlapping nperseg chunks, last one zero padded, fp32 (ceil(N/nperseg), nperseg, 2)."""
    seq = np.asarray(sequence)
    n = -(-len(seq) // nperseg)
    out = np.zeros((n, nperseg, seq.shape[1]), dtype=seq.dtype)
    for i in range(n):
        part = seq[i * nperseg:(i + 1) * nperseg]
        out[i, :len(part)] = part
    return torch.from_numpy(out.astype(np.float32))


class IQFrameDataset(Dataset):
    def __init__(self, features, targets, frame_length, stride=1):
        self.features, self.targets = frames(features, frame_length, stride), frames(targets, frame_length, stride)

    def __len__(self):
        return len(self.features)

    def __getitem__(self, i):
        return self.features[i], self.targets[i]


class IQSegmentDataset(Dataset):
    def __init__(self, features, targets, nperseg=16384):
        self.nperseg = nperseg
        self.features, self.targets = segments(features, nperseg), segments(targets, nperseg)

    def __len__(self):
        return len(self.features)

    def __getitem__(self, i):
        return self.features[i], self.targets[i]
