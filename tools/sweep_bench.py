#!/usr/bin/env python3
"""Wall time of K lockstep train_pa runs (opendpd_amd.train_pa_sweep) against ONE solo run, at the reference's batch size (256 x 200, DGRU H13)
on the bundled DPA_200MHz data (tests/golden/dpa200_dataset.npz).   python tools/sweep_bench.py [K ...]      EXP_EPOCHS, EXP_BB, EXP_H, EXP_BATCH, EXP_T"""
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

wd = tempfile.mkdtemp(prefix="odpd_sweep_")
d = dict(np.load(os.path.join(ROOT, "tests", "golden", "dpa200_dataset.npz")))
ds = os.path.join(wd, "datasets", "DPA_200MHz")
os.makedirs(ds)
open(os.path.join(ds, "spec.json"), "w").write(str(d.pop("spec")))
for k, v in d.items():
    pd.DataFrame(v, columns=["I", "Q"]).to_csv(os.path.join(ds, f"{k}.csv"), index=False)
os.environ["OPENDPD_DATASETS"] = os.path.join(wd, "datasets")
os.chdir(wd)
E = int(os.environ.get("EXP_EPOCHS", "10"))
kw = dict(dataset_name="DPA_200MHz", PA_backbone=os.environ.get("EXP_BB", "dgru"), PA_hidden_size=int(os.environ.get("EXP_H", "13")), n_epochs=E,
          batch_size=int(os.environ.get("EXP_BATCH", "256")), frame_length=int(os.environ.get("EXP_T", "200")), accelerator="cuda")


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, r


od.train_pa(seed=99, **dict(kw, n_epochs=1))          # warm the process (library load, first-launch costs)
t1, _ = timed(lambda: od.train_pa(seed=0, **kw))
print(f"solo run: {t1:.3f} s for {E} epochs ({kw['PA_backbone']} H{kw['PA_hidden_size']}, batch {kw['batch_size']} x {kw['frame_length']})", flush=True)
for K in [int(a) for a in sys.argv[1:]] or [8, 32]:
    for exact in (True, False):
        tk, res = timed(lambda: od.train_pa_sweep(seeds=tuple(range(100, 100 + K)), exact=exact, **kw))
        assert K == 1 or all(r["lockstep"] for r in res)
        print(f"{K:3d} runs in lockstep ({res[0]['mode']:5s}): {tk:.3f} s = {tk / t1:.2f} x one run ({K * t1 / tk:.1f} x the throughput of K solo runs back to back)", flush=True)
if os.environ.get("EXP_PROFILE"):            # where the host time of a throughput-mode sweep goes
    import cProfile
    import pstats
    K = int(os.environ["EXP_PROFILE"])
    pr = cProfile.Profile()
    pr.enable()
    od.train_pa_sweep(seeds=tuple(range(100, 100 + K)), exact=False, **kw)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
