"""Helpers shared by the parity tests: load a golden fixture and flatten its state dict."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture:
    """One tests/golden/<name>.npz produced by oracle/gen_golden.py from the reference."""

    def __init__(self, name):
        self.name = name
        self.d = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
        self.meta = json.loads(str(self.d["meta"]))

    def keys(self, prefix):
        """Parameter names (state-dict order) stored under '<prefix>/'."""
        return [k[len(prefix) + 1:] for k in self.d if k.startswith(prefix + "/")]

    def param_names(self, sub=""):
        """State-dict keys that are parameters (have a gradient or are frozen weights), optional sub-model."""
        return [k for k in self.keys("sd") if k.startswith(sub)]

    def flat(self, prefix, names=None, strip=""):
        names = names if names is not None else self.keys(prefix)
        return np.concatenate([self.d[f"{prefix}/{strip}{k}"].reshape(-1) for k in names]).astype(np.float32)

    def sizes(self, names, prefix="sd", strip=""):
        return [int(self.d[f"{prefix}/{strip}{k}"].size) for k in names]

    def __getitem__(self, k):
        return self.d[k]

    def __contains__(self, k):
        return k in self.d


def rel_err(a, b):
    """max |a-b| / max(|b|) — scale-relative max error."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))
