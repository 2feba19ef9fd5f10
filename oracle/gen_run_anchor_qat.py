#!/usr/bin/env python3
"""End-to-end anchor for BASELINE config 5 (TEST INFRASTRUCTURE — build container only): runs the REFERENCE (CPU) for
    main.py --step train_pa  --dataset_name APA_200MHz --PA_backbone dgru --PA_hidden_size 23 --frame_length 200 --batch_size 256
    main.py --step train_dpd ... --DPD_backbone qgru --DPD_hidden_size 10 --quant --n_bits_w 8 --n_bits_a 8 --batch_size 64
(seed 0, 1 epoch each) and stores the logged row + the quantised DPD state dict in tests/golden/ref_runs_qat.{json,npz}.
The train_dpd step runs in-process behind the harness-side bridge for the reference's import defect (quant/__init__ does not
export Sqrt / Pow).  Usage: python oracle/gen_run_anchor_qat.py"""
import glob
import json
import os
import subprocess
import tempfile

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
C = ["--dataset_name", "APA_200MHz", "--accelerator", "cpu", "--frame_length", "200", "--seed", "0", "--n_epochs", "1",
     "--PA_backbone", "dgru", "--PA_hidden_size", "23"]
Q = ["--DPD_backbone", "qgru", "--DPD_hidden_size", "10", "--quant", "--n_bits_w", "8", "--n_bits_a", "8", "--batch_size", "64"]
RUNNER = """
sys.path.insert(0, %r)
sys.dont_write_bytecode = True
import quant
from quant.modules.ops import Sqrt, Pow
quant.Sqrt, quant.Pow = Sqrt, Pow
from steps import train_dpd
from project import Project
train_dpd.main(Project())
""" % REF


def main():
    import torch
    env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", "train_pa", "--batch_size", "256"] + C, cwd=tmp, env=env,
                              stdout=subprocess.DEVNULL)
        open(os.path.join(tmp, "_runner.py"), "w").write(RUNNER)
        subprocess.check_call(["python", "_runner.py", "--step", "train_dpd"] + C + Q, cwd=tmp, env=env, stdout=subprocess.DEVNULL)
        hist = glob.glob(f"{tmp}/log/APA_200MHz/train_dpd/**/history/*.csv", recursive=True)[0]
        dpd_path = glob.glob(f"{tmp}/save/APA_200MHz/train_dpd/**/*.pt", recursive=True)[0]
        pa_path = glob.glob(f"{tmp}/save/APA_200MHz/train_pa/*.pt")[0]
        out = {"hist": pd.read_csv(hist).to_dict(orient="list"), "hist_path": os.path.relpath(hist, tmp),
               "dpd_model": os.path.relpath(dpd_path, tmp), "pa_model": os.path.relpath(pa_path, tmp), "cmd": " ".join(C + Q)}
        json.dump(out, open(os.path.join(OUT, "ref_runs_qat.json"), "w"), indent=1)
        dp = torch.load(dpd_path)
        np.savez_compressed(os.path.join(OUT, "ref_runs_qat_models.npz"), **{"dpd/" + k: v.numpy() for k, v in dp.items()})
        print(json.dumps({k: out["hist"][k] for k in ("TRAIN_LOSS", "VAL_NMSE", "VAL_ACLR_AVG", "N_PARAM")}), out["dpd_model"])


if __name__ == "__main__":
    main()
