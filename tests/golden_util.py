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


def apnrru_noise_mask(hidden):
    """APNRRU (apnrru.py:77-99) feeds its cell the raw sample rotated by its own conjugate phase: the imaginary part of that
    feature is 0 up to rounding (~1e-9), so the gradient of W_u's column 7 is rounding noise (~1e-10) and AdamW — which normalises
    every gradient by its own magnitude — turns that noise into full-size steps.  Returns the flat-parameter mask of everything
    BUT that column: optimiser trajectories are compared on it."""
    n = 2 * hidden + 3
    m = np.ones(343 + 70 * hidden, dtype=bool)
    o_wu = 96 + 1 + n
    m[o_wu + 7:o_wu + 16 * (8 + n):8 + n] = False
    return m
