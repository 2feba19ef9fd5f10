#!/usr/bin/env python3
"""End-to-end anchors for the GMP backbone (TEST INFRASTRUCTURE — build container only): RUNS the reference on CPU in a scratch
directory and stores what it logged / saved:
  * main.py --step train_pa  --PA_backbone gmp --PA_hidden_size 11 --frame_length 50 --batch_size 64 --lr 5e-3 --n_epochs 2
        -> tests/golden/ref_runs_gmp.json["train_pa_hist"], ref_runs_gmp_model.npz
  * the classical use, GMP as the pre-distorter of a neural PA model:
    main.py --step train_pa (gru H11, as oracle/gen_run_anchors.py), then
    main.py --step train_dpd --DPD_backbone gmp --DPD_hidden_size 11 (1 epoch, lr 5e-3) and main.py --step run_dpd
        -> ref_runs_gmp.json["train_dpd_hist"], ref_runs_gmp_dpd.npz (PA weights used, DPD weights reached, dpd_out CSV)
The dataset fixture is dpa200_dataset.npz (oracle/gen_run_anchors.py).  Usage: python oracle/gen_run_anchor_gmp.py"""
import glob
import json
import os
import subprocess
import tempfile

import numpy as np
import pandas as pd
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
BASE = ["--dataset_name", "DPA_200MHz", "--accelerator", "cpu", "--frame_length", "50", "--batch_size", "64", "--seed", "0"]
PA_GMP = ["--PA_backbone", "gmp", "--PA_hidden_size", "11", "--lr", "5e-3", "--n_epochs", "2"]
PA_GRU = ["--PA_backbone", "gru", "--PA_hidden_size", "11", "--lr", "1e-3"]
DPD_GMP = ["--DPD_backbone", "gmp", "--DPD_hidden_size", "11"]


def run(cwd, step, extra):
    env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
    subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", step] + BASE + extra, cwd=cwd, env=env,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        run(tmp, "train_pa", PA_GMP)
        model = glob.glob(f"{tmp}/save/DPA_200MHz/train_pa/*.pt")[0]
        hist = pd.read_csv(glob.glob(f"{tmp}/log/DPA_200MHz/train_pa/history/*.csv")[0])
        out.update(train_pa_hist=hist.to_dict(orient="list"), model=os.path.relpath(model, tmp), cmd=" ".join(BASE + PA_GMP))
        np.savez_compressed(os.path.join(OUT, "ref_runs_gmp_model.npz"), **{k: v.numpy() for k, v in torch.load(model).items()})
        print(hist.to_string())
    with tempfile.TemporaryDirectory() as tmp:
        run(tmp, "train_pa", PA_GRU + ["--n_epochs", "2"])
        run(tmp, "train_dpd", PA_GRU[:4] + DPD_GMP + ["--lr", "5e-3", "--n_epochs", "1"])
        run(tmp, "run_dpd", PA_GRU[:4] + DPD_GMP)
        paths = {"pa_model": glob.glob(f"{tmp}/save/DPA_200MHz/train_pa/*.pt")[0],
                 "dpd_model": glob.glob(f"{tmp}/save/DPA_200MHz/train_dpd/*/*.pt")[0],
                 "dpd_out": glob.glob(f"{tmp}/dpd_out/*.csv")[0]}
        hist = pd.read_csv(glob.glob(f"{tmp}/log/DPA_200MHz/train_dpd/*/history/*.csv")[0])
        out.update(train_dpd_hist=hist.to_dict(orient="list"), paths={k: os.path.relpath(v, tmp) for k, v in paths.items()},
                   cmd_dpd=" ".join(BASE + PA_GRU[:4] + DPD_GMP + ["--lr", "5e-3", "--n_epochs", "1"]))
        pa, dpd = torch.load(paths["pa_model"]), torch.load(paths["dpd_model"])
        np.savez_compressed(os.path.join(OUT, "ref_runs_gmp_dpd.npz"), **{"pa/" + k: v.numpy() for k, v in pa.items()},
                            **{"dpd/" + k: v.numpy() for k, v in dpd.items()},
                            dpd_out=pd.read_csv(paths["dpd_out"]).to_numpy().astype(np.float64))
        print(hist.to_string())
    json.dump(out, open(os.path.join(OUT, "ref_runs_gmp.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
