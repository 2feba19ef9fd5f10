#!/usr/bin/env python3
"""LR-schedule anchor (TEST INFRASTRUCTURE — build container only): RUNS the reference on CPU,
    main.py --step train_pa --dataset_name DPA_200MHz --PA_backbone gru --PA_hidden_size 11 --frame_length 50 --batch_size 64
            --lr 5e-2 --lr_schedule 1 --patience 0 --decay_factor 0.5 --lr_end 1e-3 --n_epochs 8 --seed 0
(a large learning rate and zero patience so that ReduceLROnPlateau on VAL NMSE — negative dB values — fires within a few epochs)
and stores the rows it logged: tests/golden/ref_runs_lrsched.json.  Usage: python oracle/gen_run_anchor_lrsched.py"""
import glob
import json
import os
import subprocess
import tempfile

import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ARGS = ["--dataset_name", "DPA_200MHz", "--accelerator", "cpu", "--PA_backbone", "gru", "--PA_hidden_size", "11", "--frame_length", "50",
        "--batch_size", "64", "--lr", "5e-2", "--lr_schedule", "1", "--patience", "0", "--decay_factor", "0.5", "--lr_end", "1e-3",
        "--n_epochs", "8", "--seed", "0"]


def main():
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
        subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", "train_pa"] + ARGS, cwd=tmp, env=env,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        hist = pd.read_csv(glob.glob(f"{tmp}/log/DPA_200MHz/train_pa/history/*.csv")[0])
        json.dump({"hist": hist.to_dict(orient="list"), "cmd": " ".join(ARGS)}, open(os.path.join(OUT, "ref_runs_lrsched.json"), "w"), indent=1)
        print(hist[["LR", "TRAIN_LOSS", "VAL_NMSE"]].to_string())


if __name__ == "__main__":
    main()
