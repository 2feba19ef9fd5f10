#!/usr/bin/env python3
"""Anchors for the training options no other anchor exercises (TEST INFRASTRUCTURE — build container only): RUNS the reference on
CPU, two train_pa epochs of gru H11 on DPA_200MHz (frame 50, batch 64, seed 0) with
    l1     --loss_type l1 --lr 1e-3                 (nn.L1Loss through the fused kernels' L1 branch)
    clip   --grad_clip_val 0.02 --lr 1e-3           (clip_grad_norm_ really clipping: the default 200 never does)
    noclip --grad_clip_val 0 --lr 1e-3              (train_funcs.py:41: clipping skipped)
    sgd    --opt_type sgd --lr 1e-2                 (torch.optim.SGD(momentum 0.9): the fused HIP optimiser's SGD / Adam / RMSprop kinds since r02)
    adam   --opt_type adam --lr 1e-3
    rmsprop --opt_type rmsprop --lr 1e-3
    stride7  dgru H8, --frame_length 37 --frame_stride 7 --batch_size 100   (strided frames addressed in place by the native epoch loop)
    layers2  gru H8 --PA_num_layers 2      hidden40  dgru H40             (until r04 beyond the kernels' envelope; now on the lane-per-unit / two-layer kernels)
    lstm_layers2, dgru_layers2, gru_h48, lstm_h48, vdlstm_h40, deltagru_h40 (thx .01, thh .05)   (r04: the other configurations of those kernels)
-> tests/golden/ref_runs_variants.json.  Usage: python oracle/gen_run_anchors_variants.py"""
import glob
import json
import os
import subprocess
import tempfile

import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
BASE = ["--dataset_name", "DPA_200MHz", "--accelerator", "cpu", "--PA_backbone", "gru", "--PA_hidden_size", "11", "--frame_length", "50",
        "--batch_size", "64", "--seed", "0", "--n_epochs", "2"]
# cases that replace parts of BASE (later flags win in argparse)
CASES2 = {"stride7": ["--PA_backbone", "dgru", "--PA_hidden_size", "8", "--frame_length", "37", "--frame_stride", "7", "--batch_size", "100", "--lr", "2e-3"],
          "layers2": ["--PA_backbone", "gru", "--PA_hidden_size", "8", "--PA_num_layers", "2", "--lr", "2e-3"],
          "hidden40": ["--PA_backbone", "dgru", "--PA_hidden_size", "40", "--lr", "1e-3"],
          # r04: more configurations that moved from the ATen restatements onto kernels (lane-per-unit and two-layer kernels)
          "lstm_layers2": ["--PA_backbone", "lstm", "--PA_hidden_size", "10", "--PA_num_layers", "2", "--lr", "2e-3"],
          "dgru_layers2": ["--PA_backbone", "dgru", "--PA_hidden_size", "9", "--PA_num_layers", "2", "--lr", "2e-3"],
          "gru_h48": ["--PA_backbone", "gru", "--PA_hidden_size", "48", "--lr", "1e-3"],
          "lstm_h48": ["--PA_backbone", "lstm", "--PA_hidden_size", "48", "--lr", "1e-3"],
          "vdlstm_h40": ["--PA_backbone", "vdlstm", "--PA_hidden_size", "40", "--lr", "1e-3"],
          "deltagru_h40": ["--PA_backbone", "deltagru", "--PA_hidden_size", "40", "--thx", "0.01", "--thh", "0.05", "--lr", "1e-3"]}
CASES = {"l1": ["--loss_type", "l1", "--lr", "1e-3"], "clip": ["--grad_clip_val", "0.02", "--lr", "1e-3"],
         "noclip": ["--grad_clip_val", "0", "--lr", "1e-3"], "sgd": ["--opt_type", "sgd", "--lr", "1e-2"],
         "adam": ["--opt_type", "adam", "--lr", "1e-3"], "rmsprop": ["--opt_type", "rmsprop", "--lr", "1e-3"]}


def main():
    import sys
    path = os.path.join(OUT, "ref_runs_variants.json")
    only = sys.argv[1:]                                   # names to (re)generate; the others keep their stored rows
    out = json.load(open(path)) if only and os.path.exists(path) else {}
    for name, extra in list(CASES.items()) + list(CASES2.items()):
        if only and name not in only:
            continue
        with tempfile.TemporaryDirectory() as tmp:
            env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
            subprocess.check_call(["python", os.path.join(REF, "main.py"), "--step", "train_pa"] + BASE + extra, cwd=tmp, env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            hist = pd.read_csv(glob.glob(f"{tmp}/log/DPA_200MHz/train_pa/history/*.csv")[0])
            out[name] = {"hist": hist.to_dict(orient="list"), "cmd": " ".join(BASE + extra)}
            print(name, hist[["TRAIN_LOSS", "VAL_NMSE", "TEST_ACLR_AVG"]].to_numpy().tolist())
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
