#!/usr/bin/env python3
"""End-to-end anchors for the backbones that had none of their own (TEST INFRASTRUCTURE — build container only): RUNS the
reference on CPU in a scratch directory,
    main.py --step train_pa --dataset_name DPA_200MHz --PA_backbone <bb> --PA_hidden_size <H> --frame_length 50 --batch_size 64
            --lr 2e-3 --n_epochs 2 --seed 0 [--thx 0.01 --thh 0.05]
for lstm H14, tcnn C35 and deltagru H15 (thresholded), and stores the history rows it logged (tests/golden/ref_runs_more.json).
gru, dgru, vdlstm, deltagru_tcnskip, the QAT qgru and gmp have anchors of their own (gen_run_anchors*.py).  qgru H10, qgru_amp1 H10
(float) and pgjanet H11 cannot be reached through the reference's CLI as it stands (SURVEY §0 defects 1 and 2): for them the same
`steps.train_pa.main(Project())` runs in-process behind a harness-side bridge (quant.Sqrt / quant.Pow exported; PGJANET's
constructor made to accept the `window_size=` keyword the registry passes) — the reference's files stay untouched.
Dataset fixture: dpa200_dataset.npz.
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
BRIDGED = {"qgru": ["--PA_backbone", "qgru", "--PA_hidden_size", "10"],
           "qgru_amp1": ["--PA_backbone", "qgru_amp1", "--PA_hidden_size", "10"],
           "pgjanet": ["--PA_backbone", "pgjanet", "--PA_hidden_size", "11"]}
RUNNER = """
import sys
sys.path.insert(0, %r)
sys.dont_write_bytecode = True
import quant
from quant.modules.ops import Sqrt, Pow
quant.Sqrt, quant.Pow = Sqrt, Pow                     # defect 2: quant/__init__ does not export them
import backbones.pgjanet as pj
_init = pj.PGJANET.__init__
pj.PGJANET.__init__ = lambda self, hidden_size, output_size, bias=True, window_size=None: _init(self, hidden_size, output_size, bias)  # defect 1
from steps import train_pa
from project import Project
train_pa.main(Project())
""" % REF
CASES = {"lstm": ["--PA_backbone", "lstm", "--PA_hidden_size", "14"],
         "tcnn": ["--PA_backbone", "tcnn", "--PA_hidden_size", "35"],
         "deltagru": ["--PA_backbone", "deltagru", "--PA_hidden_size", "15", "--thx", "0.01", "--thh", "0.05"]}


def main():
    out = {}
    for name, extra in list(CASES.items()) + list(BRIDGED.items()):
        with tempfile.TemporaryDirectory() as tmp:
            env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
            script = os.path.join(REF, "main.py")
            if name in BRIDGED:
                script = os.path.join(tmp, "_runner.py")
                open(script, "w").write(RUNNER)
            subprocess.check_call(["python", script, "--step", "train_pa"] + BASE + extra, cwd=tmp, env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            model = glob.glob(f"{tmp}/save/DPA_200MHz/train_pa/*.pt")[0]
            hist = pd.read_csv(glob.glob(f"{tmp}/log/DPA_200MHz/train_pa/history/*.csv")[0])
            out[name] = {"hist": hist.to_dict(orient="list"), "model": os.path.relpath(model, tmp), "cmd": " ".join(BASE + extra)}
            print(name, os.path.basename(model))
            print(hist[["TRAIN_LOSS", "VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_NMSE"]].to_string())
    json.dump(out, open(os.path.join(OUT, "ref_runs_more.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
