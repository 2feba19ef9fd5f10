#!/usr/bin/env python3
"""End-to-end anchor for `--pretrained_model` in the quantised flow (TEST INFRASTRUCTURE — build container only): runs the REFERENCE
(CPU) for
    main.py --step train_dpd --dataset_name DPA_200MHz --DPD_backbone qgru --DPD_hidden_size 10 --quant --n_bits_w 8 --n_bits_a 8
            --quant_dir_label w8a8pre --pretrained_model pygru.pt   (1 epoch, frame 50, batch 64, lr 1e-3, seed 0)
in front of the GRU H11 PA of tests/golden/ref_runs_qat_dpa.npz.  `pygru.pt` is a float checkpoint with the key names of the float
holder Base_GRUQuantEnv.load_model strict-loads into (quant_envs.py:173-182): seeded random weights and biases — the weights must
survive quantisation, the biases must be re-drawn.  Stores the logged row, the pretrained dict and the saved quantised state dict in
tests/golden/ref_runs_qat_pre.{json,npz}.  Usage: python oracle/gen_run_anchor_qat_pretrained.py"""
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
     "--PA_backbone", "gru", "--PA_hidden_size", "11", "--n_epochs", "1"]
Q = ["--DPD_backbone", "qgru", "--DPD_hidden_size", "10", "--quant", "--n_bits_w", "8", "--n_bits_a", "8", "--quant_dir_label", "w8a8pre",
     "--pretrained_model", "pygru.pt"]
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
H = 10


def main():
    import torch
    env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
    base = dict(np.load(os.path.join(OUT, "ref_runs_qat_dpa.npz")))
    prev = json.load(open(os.path.join(OUT, "ref_runs_qat_dpa.json")))
    g = torch.Generator().manual_seed(11)
    cell = "backbone.rnn.rnn_cell_list.0."
    shapes = {cell + "x2h.weight": (3 * H, 4), cell + "x2h.bias": (3 * H,), cell + "h2h.weight": (3 * H, H), cell + "h2h.bias": (3 * H,),
              "backbone.fc_out.weight": (2, H), "backbone.fc_out.bias": (2,)}
    pre = {k: (torch.rand(s, generator=g) - 0.5) * 0.6 for k, s in shapes.items()}
    with tempfile.TemporaryDirectory() as tmp:
        pa_path = os.path.join(tmp, prev["pa_model"])
        os.makedirs(os.path.dirname(pa_path), exist_ok=True)
        torch.save({k[3:]: torch.from_numpy(v) for k, v in base.items() if k.startswith("pa/")}, pa_path)
        torch.save(pre, os.path.join(tmp, "pygru.pt"))
        open(os.path.join(tmp, "_runner.py"), "w").write(RUNNER)
        log = subprocess.run(["python", "_runner.py", "--step", "train_dpd"] + C + Q, cwd=tmp, env=env, capture_output=True, text=True)
        assert log.returncode == 0, log.stderr[-2000:]
        assert "Load pretrained model from pygru.pt" in log.stdout and "Quantization setup failed" not in log.stdout
        hist = glob.glob(f"{tmp}/log/DPA_200MHz/train_dpd/**/history/*.csv", recursive=True)[0]
        dpd_path = glob.glob(f"{tmp}/save/DPA_200MHz/train_dpd/**/*.pt", recursive=True)[0]
        out = {"hist": pd.read_csv(hist).to_dict(orient="list"), "hist_path": os.path.relpath(hist, tmp),
               "dpd_model": os.path.relpath(dpd_path, tmp), "pa_model": prev["pa_model"], "cmd": " ".join(C + Q)}
        json.dump(out, open(os.path.join(OUT, "ref_runs_qat_pre.json"), "w"), indent=1)
        np.savez_compressed(os.path.join(OUT, "ref_runs_qat_pre.npz"), **{"pre/" + k: v.numpy() for k, v in pre.items()},
                            **{"dpd/" + k: v.numpy() for k, v in torch.load(dpd_path).items()})
        print(json.dumps({k: out["hist"][k] for k in ("TRAIN_LOSS", "VAL_NMSE", "VAL_ACLR_AVG", "N_PARAM")}), out["dpd_model"])


if __name__ == "__main__":
    main()
