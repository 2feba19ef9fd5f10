#!/usr/bin/env python3
"""Host-side profile of a train_dpd run through opendpd_amd.api (BASELINE config 3 shape on the bundled DPA_200MHz data:
TRes-DeltaGRU15 DPD -> frozen DGRU23 PA).  usage (GPU box): python tools/e2e_profile_dpd.py [epochs]"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pa = dict(dataset_name="DPA_200MHz", PA_backbone="dgru", PA_hidden_size=23, frame_length=200, batch_size=64, lr=1e-3, seed=0, accelerator="cuda")
od.train_pa(n_epochs=2, **pa)
kw = dict(pa, DPD_backbone="deltagru_tcnskip", DPD_hidden_size=15, thx=0.01, thh=0.05)
if len(sys.argv) > 2 and sys.argv[2] == "gru":      # GRU-family pair: the one-launch cascade step + native epoch loop
    kw = dict(pa, DPD_backbone="dgru", DPD_hidden_size=13)
od.train_dpd(n_epochs=1, **kw)   # warm-up
pr = cProfile.Profile()
t0 = time.time(); pr.enable()
od.train_dpd(n_epochs=n, **kw)
pr.disable(); t1 = time.time()
print(f"{n} epochs: {t1 - t0:.3f} s wall = {1e3 * (t1 - t0) / n:.1f} ms per epoch (train steps of 64 x 200 + val + test eval)")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30); print(s.getvalue()[:7000])
