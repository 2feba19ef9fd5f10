#!/usr/bin/env python3
"""End-to-end anchor for the GMP backbone (TEST INFRASTRUCTURE — build container only): RUNS the reference on CPU in a scratch
directory,  main.py --step train_pa --dataset_name DPA_200MHz --PA_backbone gmp --PA_hidden_size 11 --frame_length 50
--batch_size 64 --lr 5e-3 --n_epochs 2 --seed 0,  and stores the history rows it logged (tests/golden/ref_runs_gmp.json) and
the weights it saved (ref_runs_gmp_model.npz).  The dataset fixture is dpa200_dataset.npz (oracle/gen_run_anchors.py).
Usage: python oracle/gen_run_anchor_gmp.py"""
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
ARGS = ["--dataset_name", "DPA_200MHz", "--accelerator", "cpu", "--PA_backbone", "gmp", "--PA_hidden_size", "11",
        "--frame_length", "50", "--batch_size", "64", "--lr", "5e-3", "--seed", "0", "--n_epochs", "2"]


def main():
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
        subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", "train_pa"] + ARGS, cwd=tmp, env=env,
                              stdout=subprocess.DEVNULL)
        model = glob.glob(f"{tmp}/save/DPA_200MHz/train_pa/*.pt")[0]
        hist = pd.read_csv(glob.glob(f"{tmp}/log/DPA_200MHz/train_pa/history/*.csv")[0])
        json.dump({"train_pa_hist": hist.to_dict(orient="list"), "model": os.path.relpath(model, tmp), "cmd": " ".join(ARGS)},
                  open(os.path.join(OUT, "ref_runs_gmp.json"), "w"), indent=1)
        np.savez_compressed(os.path.join(OUT, "ref_runs_gmp_model.npz"), **{k: v.numpy() for k, v in torch.load(model).items()})
        print(hist.to_string())


if __name__ == "__main__":
    main()
