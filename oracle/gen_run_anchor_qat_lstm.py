#!/usr/bin/env python3
"""End-to-end anchor for `--quant` on a backbone where the surgery swaps only nn.Linear heads (TEST INFRASTRUCTURE — build container only):
runs the REFERENCE (CPU) for
    main.py --step train_dpd --dataset_name DPA_200MHz --PA_backbone gru --PA_hidden_size 11 --frame_length 50 --batch_size 64 --lr 1e-3
            --DPD_backbone lstm --DPD_hidden_size 12 --quant --n_bits_w 8 --n_bits_a 8 --quant_dir_label w8a8 (2 epochs)
    ... --step run_dpd   (same flags)
in front of the GRU PA the reference trained for tests/golden/ref_runs_qat_dpa.npz (written where train_dpd looks for it), and stores the
logged rows, the saved state dict (float nn.LSTM + INT_Linear fc_out: parameters AND side-effect buffers) and the exported CSV in
tests/golden/ref_runs_qat_lstm.{json,npz}.   Usage: python oracle/gen_run_anchor_qat_lstm.py [backbone]
(`neuraltx` / `rvtdcnn` / `pgjanet`: the same run with that --DPD_backbone -> ref_runs_qat_<backbone>.{json,npz}; pgjanet through the
harness-side bridge of the reference's constructor defect)"""
import glob
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_run_anchor_qat_dpa import C, OUT, REF, RUNNER  # noqa: E402

BB = sys.argv[1] if len(sys.argv) > 1 else "lstm"
if BB == "pgjanet":      # the reference's CoreModel hands PGJANET a keyword it does not take (SURVEY §0 defect 1): bridged harness-side, as
    # oracle/gen_run_anchors_more.py does for the float run
    RUNNER = RUNNER.replace("import importlib\n", "import importlib\nimport backbones.pgjanet as pj\n_init = pj.PGJANET.__init__\n"
                            "pj.PGJANET.__init__ = lambda self, hidden_size, output_size, bias=True, window_size=None: _init(self, hidden_size, output_size, bias)\n")
Q = ["--DPD_backbone", BB, "--DPD_hidden_size", "12", "--quant", "--n_bits_w", "8", "--n_bits_a", "8", "--quant_dir_label", "w8a8"]


def main():
    import torch
    env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
    pa = {k[3:]: torch.from_numpy(v) for k, v in np.load(os.path.join(OUT, "ref_runs_qat_dpa.npz")).items() if k.startswith("pa/")}
    pa_rel = json.load(open(os.path.join(OUT, "ref_runs_qat_dpa.json")))["pa_model"]
    with tempfile.TemporaryDirectory() as tmp:
        quiet = dict(stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.makedirs(os.path.join(tmp, os.path.dirname(pa_rel)), exist_ok=True)
        torch.save(pa, os.path.join(tmp, pa_rel))
        open(os.path.join(tmp, "_runner.py"), "w").write(RUNNER)
        subprocess.check_call(["python", "_runner.py", "--step", "train_dpd"] + C + Q, cwd=tmp, env=env, **quiet)
        subprocess.check_call(["python", "_runner.py", "--step", "run_dpd"] + C + Q, cwd=tmp, env=env, **quiet)
        hist = glob.glob(f"{tmp}/log/DPA_200MHz/train_dpd/**/history/*.csv", recursive=True)[0]
        dpd_path = glob.glob(f"{tmp}/save/DPA_200MHz/train_dpd/**/*.pt", recursive=True)[0]
        csv = glob.glob(f"{tmp}/dpd_out/**/*.csv", recursive=True)[0]
        out = {"hist": pd.read_csv(hist).to_dict(orient="list"), "hist_path": os.path.relpath(hist, tmp),
               "dpd_model": os.path.relpath(dpd_path, tmp), "pa_model": pa_rel, "dpd_out": os.path.relpath(csv, tmp), "cmd": " ".join(C + Q)}
        json.dump(out, open(os.path.join(OUT, f"ref_runs_qat_{BB}.json"), "w"), indent=1)
        np.savez_compressed(os.path.join(OUT, f"ref_runs_qat_{BB}.npz"), **{"dpd/" + k: v.numpy() for k, v in torch.load(dpd_path).items()},
                            dpd_out=pd.read_csv(csv).to_numpy().astype(np.float64))
        print(json.dumps({k: out["hist"][k] for k in ("TRAIN_LOSS", "VAL_NMSE", "VAL_ACLR_AVG", "N_PARAM")}), out["dpd_model"], out["dpd_out"])


if __name__ == "__main__":
    main()
