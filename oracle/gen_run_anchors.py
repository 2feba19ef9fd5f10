#!/usr/bin/env python3
"""Generates the end-to-end anchors of tests/golden (TEST INFRASTRUCTURE — build container only):
  * dpa200_dataset.npz     the bundled DPA_200MHz dataset (six CSV splits + spec.json) as arrays: input DATA of the run
  * ref_runs.json          history CSV rows the reference logs for
                             main.py --step train_pa  (gru H11, F50, b64, lr 1e-3, 2 epochs, seed 0)
                             main.py --step train_dpd (deltagru_tcnskip H15, thx .01, thh .05, 1 epoch, same PA)
  * ref_runs_models.npz    the PA / DPD state dicts those runs saved + the dpd_out CSV of main.py --step run_dpd
by RUNNING the reference (CPU) in a scratch directory.  Usage: python oracle/gen_run_anchors.py"""
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
COMMON = ["--dataset_name", "DPA_200MHz", "--accelerator", "cpu", "--PA_backbone", "gru", "--PA_hidden_size", "11",
          "--frame_length", "50", "--batch_size", "64", "--lr", "1e-3", "--seed", "0"]
DPD = ["--DPD_backbone", "deltagru_tcnskip", "--DPD_hidden_size", "15", "--thx", "0.01", "--thh", "0.05"]


def run(cwd, step, extra):
    env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
    subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", step] + COMMON + extra, cwd=cwd, env=env,
                          stdout=subprocess.DEVNULL)


def main():
    R = os.path.join(REF, "datasets", "DPA_200MHz")
    d = {f"{s}_{k}": pd.read_csv(f"{R}/{s}_{k}.csv").to_numpy() for s in ("train", "val", "test") for k in ("input", "output")}
    np.savez_compressed(os.path.join(OUT, "dpa200_dataset.npz"), spec=np.array(json.dumps(json.load(open(f"{R}/spec.json")))), **d)
    with tempfile.TemporaryDirectory() as tmp:
        run(tmp, "train_pa", ["--n_epochs", "2"])
        run(tmp, "train_dpd", DPD + ["--n_epochs", "1"])
        run(tmp, "run_dpd", DPD)
        paths = {"pa_model": glob.glob(f"{tmp}/save/DPA_200MHz/train_pa/*.pt")[0],
                 "dpd_model": glob.glob(f"{tmp}/save/DPA_200MHz/train_dpd/*/*.pt")[0],
                 "dpd_out": glob.glob(f"{tmp}/dpd_out/*.csv")[0]}
        out = {"train_pa_hist": pd.read_csv(glob.glob(f"{tmp}/log/DPA_200MHz/train_pa/history/*.csv")[0]).to_dict(orient="list"),
               "train_dpd_hist": pd.read_csv(glob.glob(f"{tmp}/log/DPA_200MHz/train_dpd/*/history/*.csv")[0]).to_dict(orient="list"),
               "paths": {k: os.path.relpath(v, tmp) for k, v in paths.items()},
               "cmd": {"common": " ".join(COMMON), "train_dpd": " ".join(DPD)}}
        json.dump(out, open(os.path.join(OUT, "ref_runs.json"), "w"), indent=1)
        pa, dpd = torch.load(paths["pa_model"]), torch.load(paths["dpd_model"])
        np.savez_compressed(os.path.join(OUT, "ref_runs_models.npz"), **{"pa/" + k: v.numpy() for k, v in pa.items()},
                            **{"dpd/" + k: v.numpy() for k, v in dpd.items()},
                            dpd_out=pd.read_csv(paths["dpd_out"]).to_numpy().astype(np.float64))


if __name__ == "__main__":
    main()
