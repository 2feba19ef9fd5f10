#!/usr/bin/env python3
"""Where the wall time of a lockstep sweep goes, phase by phase (synchronised timers around setup, train epochs, evaluation, bookkeeping).
usage: tools/sweep_phase_time.py [K]          (throughput mode, DGRU H13, 10 epochs on the bundled DPA_200MHz split, as tools/sweep_bench.py)"""
import collections
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import opendpd_amd as od  # noqa: E402
from opendpd_amd import sweep as SW, project as PJ  # noqa: E402

wd = tempfile.mkdtemp(prefix="odpd_sweep_")
d = dict(np.load(os.path.join(ROOT, "tests", "golden", "dpa200_dataset.npz")))
ds = os.path.join(wd, "datasets", "DPA_200MHz")
os.makedirs(ds)
open(os.path.join(ds, "spec.json"), "w").write(str(d.pop("spec")))
for k, v in d.items():
    pd.DataFrame(v, columns=["I", "Q"]).to_csv(os.path.join(ds, f"{k}.csv"), index=False)
os.environ["OPENDPD_DATASETS"] = os.path.join(wd, "datasets")
os.chdir(wd)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kw = dict(dataset_name="DPA_200MHz", PA_backbone="dgru", PA_hidden_size=13, n_epochs=10, batch_size=256, frame_length=200, accelerator="cuda")
acc = collections.defaultdict(float)


def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[name] += time.perf_counter() - t0
        return r
    return w


od.train_pa(seed=99, **dict(kw, n_epochs=1))
od.train_pa_sweep(seeds=(1, 2), exact=False, **dict(kw, n_epochs=1))
SW._setup = timed("setup (Project, model init, loaders, logger, optimiser)", SW._setup)
SW._Group.train_epoch = timed("train epochs (incl. the GPU)", SW._Group.train_epoch)
SW._Group.eval_epoch = timed("evaluation + metrics (incl. the GPU)", SW._Group.eval_epoch)
SW._Group.__init__ = timed("group buffers", SW._Group.__init__)
PJ.Project.finish_epoch = timed("finish_epoch (log rows, best model, scheduler)", PJ.Project.finish_epoch)
PJ.CsvLogger.flush = timed("checkpoint flush", PJ.CsvLogger.flush)
torch.cuda.synchronize()
t0 = time.perf_counter()
od.train_pa_sweep(seeds=tuple(range(100, 100 + K)), exact=False, **kw)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
for k, v in acc.items():
    print(f"{k:60s} {v * 1e3:8.1f} ms")
print(f"{'total':60s} {tot * 1e3:8.1f} ms   (accounted {sum(acc.values()) * 1e3:.1f})")
