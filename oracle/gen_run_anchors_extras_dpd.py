#!/usr/bin/env python3
"""End-to-end train_dpd anchors for the SURVEY §8 f4 backbones in the DPD role (TEST INFRASTRUCTURE — build container only): RUNS the
reference on CPU — one train_pa epoch of the PA (gru H11) on DPA_200MHz (frame 50, batch 64, lr 2e-3, seed 0), then for each DPD backbone
one train_dpd epoch through the frozen PA — and stores the logged rows: tests/golden/ref_runs_extras_dpd.json (+ the PA checkpoint the
DPD runs started from, ref_runs_extras_dpd.npz, so that the runs here start from the same PA).
Usage: python oracle/gen_run_anchors_extras_dpd.py"""
import glob
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pandas as pd
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
BASE = ["--dataset_name", "DPA_200MHz", "--accelerator", "cpu", "--frame_length", "50", "--batch_size", "64", "--seed", "0", "--lr", "2e-3",
        "--n_epochs", "1", "--PA_backbone", "gru", "--PA_hidden_size", "11"]
CASES = {"rvtdcnn": 6, "bojanet": 8, "deltajanet": 10, "dvrjanet": 8, "neuraltx": 12, "mcldnn": 8}
# `python oracle/gen_run_anchors_extras_dpd.py hot`: the hot-path backbones in the DPD role -> ref_runs_hot_dpd.{json,npz}
HOT = {"gru": 11, "dgru": 9, "lstm": 10, "vdlstm": 9, "tcnn": 20, "qgru": 10, "qgru_amp1": 10, "deltagru": 12}


def main():
    import sys
    hot = len(sys.argv) > 1 and sys.argv[1] == "hot"
    cases, stem = (HOT, "ref_runs_hot_dpd") if hot else (CASES, "ref_runs_extras_dpd")
    out = {}
    env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", "train_pa"] + BASE, cwd=tmp, env=env,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1500)
        pa = glob.glob(f"{tmp}/save/DPA_200MHz/train_pa/*.pt")[0]
        sd = torch.load(pa, map_location="cpu")
        arrays = {"pa/" + k: v.numpy() for k, v in sd.items()}
        out["pa_model"] = os.path.relpath(pa, tmp)
        for bb, H in cases.items():
            extra = ["--DPD_backbone", bb, "--DPD_hidden_size", str(H)] + (["--thx", "0.01", "--thh", "0.03"] if bb == "deltagru" else [])
            for d in ("log/DPA_200MHz/train_dpd", "save/DPA_200MHz/train_dpd"):
                shutil.rmtree(os.path.join(tmp, d), ignore_errors=True)
            try:
                subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", "train_dpd"] + BASE + extra, cwd=tmp, env=env,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=2400)
            except Exception as e:      # noqa: BLE001
                print(bb, "FAILED", e)
                continue
            hist = pd.read_csv(glob.glob(f"{tmp}/log/DPA_200MHz/train_dpd/*/history/*.csv")[0])
            model = glob.glob(f"{tmp}/save/DPA_200MHz/train_dpd/*/*.pt")[0]
            out[bb] = {"hist": hist.to_dict(orient="list"), "hidden": H, "model": os.path.relpath(model, tmp), "cmd": " ".join(BASE + extra)}
            # the trained DPD and what main.py --step run_dpd exports with it (dpd_out/<id>.csv: I, Q, I_dpd, Q_dpd)
            shutil.rmtree(os.path.join(tmp, "dpd_out"), ignore_errors=True)
            try:
                subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", "run_dpd"] + BASE + extra, cwd=tmp, env=env,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1200)
                csv = glob.glob(f"{tmp}/dpd_out/*.csv")[0]
                out[bb]["dpd_out"] = os.path.relpath(csv, tmp)
                arrays[f"{bb}/dpd_out"] = pd.read_csv(csv).to_numpy().astype(np.float64)
                for k, v in torch.load(model, map_location="cpu").items():
                    arrays[f"{bb}/sd/{k}"] = v.numpy()
            except Exception as e:      # noqa: BLE001
                print(bb, "run_dpd FAILED", e)
            print(bb, os.path.basename(model), hist.iloc[0].to_dict(), flush=True)
        np.savez_compressed(os.path.join(OUT, stem + ".npz"), **arrays)
    json.dump(out, open(os.path.join(OUT, stem + ".json"), "w"), indent=1)


if __name__ == "__main__":
    main()
