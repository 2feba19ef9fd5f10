#!/usr/bin/env python3
"""Runs the anchor commands of tests/test_e2e_gpu.py and prints our log rows next to the reference's."""
import json, os, sys, tempfile, time
import numpy as np, pandas as pd, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")
ref = json.load(open(os.path.join(G, "ref_runs.json")))
m = dict(np.load(os.path.join(G, "ref_runs_models.npz")))
wd = tempfile.mkdtemp(); os.chdir(wd)
d = dict(np.load(os.path.join(G, "dpa200_dataset.npz")))
os.makedirs("datasets/DPA_200MHz"); open("datasets/DPA_200MHz/spec.json", "w").write(str(d.pop("spec")))
for k, v in d.items():
    pd.DataFrame(v, columns=["I", "Q"]).to_csv(f"datasets/DPA_200MHz/{k}.csv", index=False)
os.environ["OPENDPD_DATASETS"] = os.path.join(wd, "datasets")
import opendpd_amd as od
t0 = time.time()
r = od.train_pa(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, frame_length=50, batch_size=64, lr=1e-3, n_epochs=2, seed=0, accelerator="cuda")
t1 = time.time()
h = pd.read_csv(glob := os.path.join("log/DPA_200MHz/train_pa/history", os.path.basename(r["log_path"])))
cols = ["TRAIN_LOSS", "VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_EVM", "TEST_ACLR_AVG"]
print(f"train_pa gru H11 F50 b64 2 epochs: {t1-t0:.2f} s wall (reference CPU: 9.7 s of epochs + setup)")
for ep in range(2):
    print(" epoch", ep, " ours:", [round(float(h[c][ep]), 6) for c in cols]); print("          ref:", [round(ref["train_pa_hist"][c][ep], 6) for c in cols])
torch.save({k[3:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("pa/")}, ref["paths"]["pa_model"])
kw = dict(dataset_name="DPA_200MHz", PA_backbone="gru", PA_hidden_size=11, DPD_backbone="deltagru_tcnskip", DPD_hidden_size=15, frame_length=50, seed=0, accelerator="cuda")
t0 = time.time(); r = od.train_dpd(batch_size=64, lr=1e-3, n_epochs=1, thx=0.01, thh=0.05, **kw); t1 = time.time()
h = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(r["log_path"])), "history", os.path.basename(r["log_path"])))
cols = ["TRAIN_LOSS", "SP_T_DX", "SP_T_DH", "HW_PARAM", "VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"]
print(f"train_dpd TRes-DeltaGRU15 -> GRU11, 1 epoch: {t1-t0:.2f} s wall (reference CPU: 16.7 s)")
print("  ours:", [round(float(h[c][0]), 6) for c in cols]); print("   ref:", [round(ref["train_dpd_hist"][c][0], 6) for c in cols])
