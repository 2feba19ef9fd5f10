#!/usr/bin/env python3
"""End-to-end anchors on the APA datasets of BASELINE configs 2 and 4 (TEST INFRASTRUCTURE — build container only):
  * apa200_dataset.npz / apa200b_dataset.npz   the bundled APA_200MHz / APA_200MHz_b splits + spec.json as arrays (input DATA)
  * ref_runs_apa.json   history CSV rows the reference logs for
        main.py --step train_pa --dataset_name APA_200MHz   --PA_backbone dgru   --PA_hidden_size 13 --frame_length 200 --batch_size 256
        main.py --step train_pa --dataset_name APA_200MHz_b --PA_backbone vdlstm --PA_hidden_size 13 --frame_length 200 --batch_size 256
    (lr 5e-4 default, seed 0, 1 epoch each), and for BASELINE config 3 on APA_200MHz (train_pa dgru H23, then train_dpd
    deltagru_tcnskip H15 thx .01 thh .05 b64: 919 steps; + ref_runs_apa_models.npz with the PA / DPD state dicts those runs saved)
    by RUNNING the reference (CPU) in a scratch directory.
Usage: python oracle/gen_run_anchors_apa.py"""
import glob
import json
import os
import subprocess
import tempfile

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
RUNS = {"dgru_apa200": ["--dataset_name", "APA_200MHz", "--PA_backbone", "dgru", "--PA_hidden_size", "13"],
        "vdlstm_apa200b": ["--dataset_name", "APA_200MHz_b", "--PA_backbone", "vdlstm", "--PA_hidden_size", "13"]}
COMMON = ["--accelerator", "cpu", "--frame_length", "200", "--batch_size", "256", "--seed", "0", "--n_epochs", "1"]


def main():
    for ds, tag in (("APA_200MHz", "apa200"), ("APA_200MHz_b", "apa200b")):
        R = os.path.join(REF, "datasets", ds)
        d = {f"{s}_{k}": pd.read_csv(f"{R}/{s}_{k}.csv").to_numpy() for s in ("train", "val", "test") for k in ("input", "output")}
        np.savez_compressed(os.path.join(OUT, f"{tag}_dataset.npz"), spec=np.array(json.dumps(json.load(open(f"{R}/spec.json")))), **d)
    out = {"cmd": {k: " ".join(v + COMMON) for k, v in RUNS.items()}}
    env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
    for k, args in RUNS.items():
        with tempfile.TemporaryDirectory() as tmp:
            subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", "train_pa"] + args + COMMON, cwd=tmp, env=env,
                                  stdout=subprocess.DEVNULL)
            hist = glob.glob(f"{tmp}/log/*/train_pa/history/*.csv")[0]
            out[k] = {"hist": pd.read_csv(hist).to_dict(orient="list"), "model_id": os.path.basename(hist)[:-4]}
    # BASELINE config 3 on APA_200MHz: train_pa dgru H23 (1 epoch) -> train_dpd TRes-DeltaGRU H15, thx .01, thh .05, b64 (1 epoch)
    import torch
    c3 = ["--dataset_name", "APA_200MHz", "--accelerator", "cpu", "--frame_length", "200", "--seed", "0", "--n_epochs", "1",
          "--PA_backbone", "dgru", "--PA_hidden_size", "23"]
    dpd = ["--DPD_backbone", "deltagru_tcnskip", "--DPD_hidden_size", "15", "--thx", "0.01", "--thh", "0.05", "--batch_size", "64"]
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", "train_pa", "--batch_size", "256"] + c3, cwd=tmp, env=env,
                              stdout=subprocess.DEVNULL)
        subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", "train_dpd"] + c3 + dpd, cwd=tmp, env=env,
                              stdout=subprocess.DEVNULL)
        pa_path = glob.glob(f"{tmp}/save/APA_200MHz/train_pa/*.pt")[0]
        dpd_path = glob.glob(f"{tmp}/save/APA_200MHz/train_dpd/*/*.pt")[0]
        hist = glob.glob(f"{tmp}/log/APA_200MHz/train_dpd/*/history/*.csv")[0]
        out["config3_apa200"] = {"hist": pd.read_csv(hist).to_dict(orient="list"), "pa_model": os.path.relpath(pa_path, tmp),
                                 "dpd_model": os.path.relpath(dpd_path, tmp), "cmd": " ".join(c3 + dpd)}
        pa, dp = torch.load(pa_path), torch.load(dpd_path)
        np.savez_compressed(os.path.join(OUT, "ref_runs_apa_models.npz"), **{"pa/" + k: v.numpy() for k, v in pa.items()},
                            **{"dpd/" + k: v.numpy() for k, v in dp.items()})
    json.dump(out, open(os.path.join(OUT, "ref_runs_apa.json"), "w"), indent=1)
    print(json.dumps({k: {c: v["hist"][c] for c in ("TRAIN_LOSS", "VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "N_PARAM")} for k, v in out.items() if k != "cmd"}))


if __name__ == "__main__":
    main()
