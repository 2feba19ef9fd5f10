#!/usr/bin/env python3
"""End-to-end anchor for the quantised flow on the small dataset (TEST INFRASTRUCTURE — build container only): runs the REFERENCE
(CPU) for
    main.py --step train_pa  --dataset_name DPA_200MHz --PA_backbone gru --PA_hidden_size 11 --frame_length 50 --batch_size 64 --lr 1e-3 (2 epochs)
    ... --step train_dpd --DPD_backbone qgru --DPD_hidden_size 10 --quant --n_bits_w 8 --n_bits_a 8 --quant_dir_label w8a8 (2 epochs)
    ... --step run_dpd   (same flags)
and stores the logged rows, the saved quantised state dict (parameters AND side-effect buffers) and the exported CSV in
tests/golden/ref_runs_qat_dpa.{json,npz}.  train_dpd / run_dpd run in-process behind the harness-side bridge for the reference's
import defect (quant/__init__ does not export Sqrt / Pow).  Usage: python oracle/gen_run_anchor_qat_dpa.py"""
import glob
import json
import os
import subprocess
import tempfile

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
C = ["--dataset_name", "DPA_200MHz", "--accelerator", "cpu", "--frame_length", "50", "--batch_size", "64", "--seed", "0", "--lr", "1e-3",
     "--PA_backbone", "gru", "--PA_hidden_size", "11", "--n_epochs", "2"]
Q = ["--DPD_backbone", "qgru", "--DPD_hidden_size", "10", "--quant", "--n_bits_w", "8", "--n_bits_a", "8", "--quant_dir_label", "w8a8"]
RUNNER = """
import sys
sys.path.insert(0, %r)
sys.dont_write_bytecode = True
import quant
from quant.modules.ops import Sqrt, Pow
quant.Sqrt, quant.Pow = Sqrt, Pow
import importlib
step = sys.argv[sys.argv.index("--step") + 1]
from project import Project
importlib.import_module("steps." + step).main(Project())
""" % REF


def main():
    import torch
    env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
    with tempfile.TemporaryDirectory() as tmp:
        quiet = dict(stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", "train_pa"] + C, cwd=tmp, env=env, **quiet)
        open(os.path.join(tmp, "_runner.py"), "w").write(RUNNER)
        subprocess.check_call(["python", "_runner.py", "--step", "train_dpd"] + C + Q, cwd=tmp, env=env, **quiet)
        subprocess.check_call(["python", "_runner.py", "--step", "run_dpd"] + C + Q, cwd=tmp, env=env, **quiet)
        hist = glob.glob(f"{tmp}/log/DPA_200MHz/train_dpd/**/history/*.csv", recursive=True)[0]
        dpd_path = glob.glob(f"{tmp}/save/DPA_200MHz/train_dpd/**/*.pt", recursive=True)[0]
        pa_path = glob.glob(f"{tmp}/save/DPA_200MHz/train_pa/*.pt")[0]
        csv = glob.glob(f"{tmp}/dpd_out/**/*.csv", recursive=True)[0]
        out = {"hist": pd.read_csv(hist).to_dict(orient="list"), "hist_path": os.path.relpath(hist, tmp),
               "dpd_model": os.path.relpath(dpd_path, tmp), "pa_model": os.path.relpath(pa_path, tmp), "dpd_out": os.path.relpath(csv, tmp),
               "cmd": " ".join(C + Q)}
        json.dump(out, open(os.path.join(OUT, "ref_runs_qat_dpa.json"), "w"), indent=1)
        np.savez_compressed(os.path.join(OUT, "ref_runs_qat_dpa.npz"), **{"dpd/" + k: v.numpy() for k, v in torch.load(dpd_path).items()},
                            **{"pa/" + k: v.numpy() for k, v in torch.load(pa_path).items()},
                            dpd_out=pd.read_csv(csv).to_numpy().astype(np.float64))
        print(json.dumps({k: out["hist"][k] for k in ("TRAIN_LOSS", "VAL_NMSE", "VAL_ACLR_AVG", "N_PARAM")}), out["dpd_model"], out["dpd_out"])


if __name__ == "__main__":
    main()
