#!/usr/bin/env python3
"""BASELINE configs 2 and 4 for one epoch through opendpd_amd.api on the GPU next to the reference's logged row
(tests/golden/ref_runs_apa.json).  usage (GPU box): python tools/e2e_compare_apa.py"""
import json, os, sys, tempfile, time
import numpy as np, pandas as pd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")
wd = tempfile.mkdtemp(); os.chdir(wd)
for tag, name in (("apa200", "APA_200MHz"), ("apa200b", "APA_200MHz_b")):
    d = dict(np.load(os.path.join(G, f"{tag}_dataset.npz")))
    os.makedirs(f"datasets/{name}"); open(f"datasets/{name}/spec.json", "w").write(str(d.pop("spec")))
    for k, v in d.items():
        pd.DataFrame(v, columns=["I", "Q"]).to_csv(f"datasets/{name}/{k}.csv", index=False)
os.environ["OPENDPD_DATASETS"] = os.path.join(wd, "datasets")
import opendpd_amd as od
ref = json.load(open(os.path.join(G, "ref_runs_apa.json")))
import torch
m = dict(np.load(os.path.join(G, "ref_runs_apa_models.npz")))
c3 = ref["config3_apa200"]
os.makedirs(os.path.dirname(c3["pa_model"]), exist_ok=True)
torch.save({k[3:]: torch.from_numpy(v) for k, v in m.items() if k.startswith("pa/")}, c3["pa_model"])
t0 = time.time()
res = od.train_dpd(dataset_name="APA_200MHz", PA_backbone="dgru", PA_hidden_size=23, DPD_backbone="deltagru_tcnskip", DPD_hidden_size=15,
                   thx=0.01, thh=0.05, frame_length=200, batch_size=64, seed=0, n_epochs=1, accelerator="cuda")
dt = time.time() - t0
h = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
print(f"config 3 (train_dpd TRes-DeltaGRU15 -> frozen reference-trained DGRU23, 919 steps of 64 x 200): {dt:.2f} s wall")
for c in ("TRAIN_LOSS", "SP_T_DX", "SP_T_DH", "VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
    print(f"   {c:14s} here {h[c][0]:.6f}   reference {c3['hist'][c][0]:.6f}")
q = json.load(open(os.path.join(G, "ref_runs_qat.json")))
t0 = time.time()
res = od.train_dpd(dataset_name="APA_200MHz", PA_backbone="dgru", PA_hidden_size=23, DPD_backbone="qgru", DPD_hidden_size=10, quant=True,
                   n_bits_w=8, n_bits_a=8, frame_length=200, batch_size=64, seed=0, n_epochs=1, accelerator="cuda")
dt = time.time() - t0
h = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(res["log_path"])), "history", os.path.basename(res["log_path"])))
print(f"config 5 (QAT W8A8 train_dpd QGRU10 -> frozen reference-trained DGRU23, 919 steps of 64 x 200): {dt:.2f} s wall")
for c in ("TRAIN_LOSS", "VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_ACLR_AVG"):
    print(f"   {c:14s} here {h[c][0]:.6f}   reference {q['hist'][c][0]:.6f}")
for key, ds, bb in (("dgru_apa200", "APA_200MHz", "dgru"), ("vdlstm_apa200b", "APA_200MHz_b", "vdlstm")):
    t0 = time.time()
    res = od.train_pa(dataset_name=ds, PA_backbone=bb, PA_hidden_size=13, frame_length=200, batch_size=256, seed=0, n_epochs=1, accelerator="cuda")
    dt = time.time() - t0
    h = pd.read_csv(os.path.join("log", ds, "train_pa", "history", os.path.basename(res["log_path"])))
    r = ref[key]["hist"]
    print(f"{key}: {dt:.2f} s wall (dataset load + 230 steps + val + test)")
    for c in ("TRAIN_LOSS", "VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE", "TEST_EVM", "TEST_ACLR_AVG"):
        print(f"   {c:14s} here {h[c][0]:.6f}   reference {r[c][0]:.6f}")
