#!/usr/bin/env python3
"""End-to-end anchors for the registry backbones that still run as torch restatements (backbones/extras.py) — TEST
INFRASTRUCTURE, build container only: RUNS the reference on CPU, one train_pa epoch each on DPA_200MHz (frame 50, batch 256,
lr 2e-3, seed 0; thx 0.01 / thh 0.05 for deltajanet), and stores the logged row: tests/golden/ref_runs_extras.json.
Usage: python oracle/gen_run_anchors_extras.py"""
import glob
import json
import os
import subprocess
import tempfile

import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
BASE = ["--dataset_name", "DPA_200MHz", "--accelerator", "cpu", "--frame_length", "50", "--batch_size", "256", "--seed", "0", "--lr", "2e-3",
        "--n_epochs", "1"]
CASES = {"rvtdcnn": 6, "apnrru": 8, "bojanet": 8, "deltajanet": 10, "dvrjanet": 8, "neuraltx": 12, "mcldnn": 8}


def main():
    out = {}
    for bb, H in CASES.items():
        extra = ["--PA_backbone", bb, "--PA_hidden_size", str(H)] + (["--thx", "0.01", "--thh", "0.05"] if bb == "deltajanet" else [])
        with tempfile.TemporaryDirectory() as tmp:
            env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
            try:
                subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", "train_pa"] + BASE + extra, cwd=tmp, env=env,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1500)
            except Exception as e:      # noqa: BLE001
                print(bb, "FAILED", e)
                continue
            hist = pd.read_csv(glob.glob(f"{tmp}/log/DPA_200MHz/train_pa/history/*.csv")[0])
            model = glob.glob(f"{tmp}/save/DPA_200MHz/train_pa/*.pt")[0]
            out[bb] = {"hist": hist.to_dict(orient="list"), "hidden": H, "model": os.path.relpath(model, tmp), "cmd": " ".join(BASE + extra)}
            print(bb, os.path.basename(model), hist[["TRAIN_LOSS", "VAL_NMSE", "TEST_ACLR_AVG"]].to_numpy().tolist(), flush=True)
    json.dump(out, open(os.path.join(OUT, "ref_runs_extras.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
