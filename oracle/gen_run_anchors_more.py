#!/usr/bin/env python3
"""End-to-end anchors for the backbones that had none of their own (TEST INFRASTRUCTURE — build container only): RUNS the
reference on CPU in a scratch directory,
    main.py --step train_pa --dataset_name DPA_200MHz --PA_backbone <bb> --PA_hidden_size <H> --frame_length 50 --batch_size 64
            --lr 2e-3 --n_epochs 2 --seed 0 [--thx 0.01 --thh 0.05]
for lstm H14, tcnn C35 and deltagru H15 (thresholded), and stores the history rows it logged (tests/golden/ref_runs_more.json).
gru, dgru, vdlstm, deltagru_tcnskip, the QAT qgru and gmp have anchors of their own (gen_run_anchors*.py); qgru / qgru_amp1 / pgjanet
cannot be reached through the reference's CLI (SURVEY §0 defects 1 and 2).  Dataset fixture: dpa200_dataset.npz.
Usage: python oracle/gen_run_anchors_more.py"""
import glob
import json
import os
import subprocess
import tempfile

import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
BASE = ["--dataset_name", "DPA_200MHz", "--accelerator", "cpu", "--frame_length", "50", "--batch_size", "64", "--seed", "0", "--lr", "2e-3",
        "--n_epochs", "2"]
CASES = {"lstm": ["--PA_backbone", "lstm", "--PA_hidden_size", "14"],
         "tcnn": ["--PA_backbone", "tcnn", "--PA_hidden_size", "35"],
         "deltagru": ["--PA_backbone", "deltagru", "--PA_hidden_size", "15", "--thx", "0.01", "--thh", "0.05"]}


def main():
    out = {}
    for name, extra in CASES.items():
        with tempfile.TemporaryDirectory() as tmp:
            env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
            subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", "train_pa"] + BASE + extra, cwd=tmp, env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            model = glob.glob(f"{tmp}/save/DPA_200MHz/train_pa/*.pt")[0]
            hist = pd.read_csv(glob.glob(f"{tmp}/log/DPA_200MHz/train_pa/history/*.csv")[0])
            out[name] = {"hist": hist.to_dict(orient="list"), "model": os.path.relpath(model, tmp), "cmd": " ".join(BASE + extra)}
            print(name, os.path.basename(model))
            print(hist[["TRAIN_LOSS", "VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE"]].to_string())
    json.dump(out, open(os.path.join(OUT, "ref_runs_more.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
