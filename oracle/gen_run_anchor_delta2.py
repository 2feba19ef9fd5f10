#!/usr/bin/env python3
"""Two-epoch train_dpd anchor for the thresholded TRes-DeltaGRU (TEST INFRASTRUCTURE — build container only): the temporal-sparsity
columns SP_T_DX / SP_T_DH / SP_T_DV / HW_PARAM are read from counters that the reference resets once per epoch (paths.py:49-59)
and that its evaluation forwards also feed; a second epoch pins that bookkeeping.  RUNS the reference on CPU:
    main.py --step train_pa  (gru H11, F50, b64, lr 1e-3, 2 epochs, seed 0, DPA_200MHz)
    main.py --step train_dpd --DPD_backbone deltagru_tcnskip --DPD_hidden_size 15 --thx 0.01 --thh 0.05 --n_epochs 2
-> tests/golden/ref_runs_delta2.json (+ the PA weights used: ref_runs_delta2_pa.npz).  Usage: python oracle/gen_run_anchor_delta2.py"""
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
          "--frame_length", "50", "--batch_size", "64", "--lr", "1e-3", "--seed", "0", "--n_epochs", "2"]
DPD = ["--DPD_backbone", "deltagru_tcnskip", "--DPD_hidden_size", "15", "--thx", "0.01", "--thh", "0.05"]


def main():
    env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
    with tempfile.TemporaryDirectory() as tmp:
        for step, extra in (("train_pa", []), ("train_dpd", DPD)):
            subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", step] + COMMON + extra, cwd=tmp, env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        pa = glob.glob(f"{tmp}/save/DPA_200MHz/train_pa/*.pt")[0]
        dpd = glob.glob(f"{tmp}/save/DPA_200MHz/train_dpd/*/*.pt")[0]
        hist = pd.read_csv(glob.glob(f"{tmp}/log/DPA_200MHz/train_dpd/*/history/*.csv")[0])
        json.dump({"hist": hist.to_dict(orient="list"), "pa_model": os.path.relpath(pa, tmp), "dpd_model": os.path.relpath(dpd, tmp),
                   "cmd": " ".join(COMMON + DPD)}, open(os.path.join(OUT, "ref_runs_delta2.json"), "w"), indent=1)
        np.savez_compressed(os.path.join(OUT, "ref_runs_delta2_pa.npz"), **{k: v.numpy() for k, v in torch.load(pa).items()})
        print(hist[["TRAIN_LOSS", "SP_T_DX", "SP_T_DH", "SP_T_DV", "HW_PARAM", "VAL_NMSE", "VAL_ACLR_AVG"]].to_string())


if __name__ == "__main__":
    main()
