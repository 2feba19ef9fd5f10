#!/usr/bin/env python3
"""Host-side profile of a train_pa run through opendpd_amd.api (where does an epoch's wall-clock go once the
kernels are fast?).  usage (GPU box): [E2E_BACKBONE=lstm E2E_HIDDEN=14] python tools/e2e_profile.py [epochs]"""
import cProfile, io, os, pstats, sys, tempfile, time
import numpy as np, pandas as pd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")
wd = tempfile.mkdtemp(); os.chdir(wd)
d = dict(np.load(os.path.join(G, "dpa200_dataset.npz")))
os.makedirs("datasets/DPA_200MHz"); open("datasets/DPA_200MHz/spec.json", "w").write(str(d.pop("spec")))
for k, v in d.items():
    pd.DataFrame(v, columns=["I", "Q"]).to_csv(f"datasets/DPA_200MHz/{k}.csv", index=False)
os.environ["OPENDPD_DATASETS"] = os.path.join(wd, "datasets")
import opendpd_amd as od
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
kw = dict(dataset_name="DPA_200MHz", PA_backbone=os.environ.get("E2E_BACKBONE", "dgru"), PA_hidden_size=int(os.environ.get("E2E_HIDDEN", "13")), frame_length=50,
          batch_size=64, lr=1e-3, seed=0, accelerator="cuda")
od.train_pa(n_epochs=1, **kw)   # warm-up: library load, first-touch
pr = cProfile.Profile()
t0 = time.time(); pr.enable()
od.train_pa(n_epochs=n, **kw)
pr.disable(); t1 = time.time()
print(f"{n} epochs: {t1 - t0:.3f} s wall = {1e3 * (t1 - t0) / n:.1f} ms per epoch (360 train steps + val + test eval)")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
