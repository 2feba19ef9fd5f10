#!/usr/bin/env python3
"""End-to-end anchor of the reference's OpenDPDv2 recipe (bash_scripts/OpenDPDv2.sh:47-117) on APA_200MHz (TEST INFRASTRUCTURE —
build container only).  Runs the REFERENCE (CPU), 1 epoch per stage, seed 0:
    main.py --step train_pa  --PA_backbone dgru --PA_hidden_size 23 --frame_length 200 --batch_size 256
    main.py --step train_dpd --DPD_backbone deltagru_tcnskip --DPD_hidden_size 15 --thx 0.01 --thh 0.05 --batch_size 64 --lr 5e-3      (float pre-training)
    main.py --step train_dpd ... --quant --n_bits_w 16 --n_bits_a 16 --quant_dir_label w16a16 --pretrained_model <float checkpoint>     (QAT stage)
    main.py --step run_dpd   ... --quant ... --quant_dir_label w16a16
and, for the second half of the surgery's envelope, one W8A8 epoch of a quantised `gru` DPD (H 11) on DPA_200MHz (frame 50).
Stores the logged rows, the PA / float-DPD / quantised-DPD state dicts and the head of the exported CSV in
tests/golden/ref_runs_v2.{json,npz}.  The quantised steps run in-process behind the harness-side bridge for the reference's import
defect (quant/__init__ does not export Sqrt / Pow).   Usage: python oracle/gen_run_anchor_opendpdv2.py"""
import glob
import json
import os
import subprocess
import tempfile

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
RUNNER = """
import sys
sys.path.insert(0, %r)
sys.dont_write_bytecode = True
import quant
from quant.modules.ops import Sqrt, Pow
quant.Sqrt, quant.Pow = Sqrt, Pow
from steps import train_dpd, run_dpd
from project import Project
proj = Project()
(train_dpd if proj.step == "train_dpd" else run_dpd).main(proj)
""" % REF


def sd_np(path, prefix):
    import torch
    return {f"{prefix}/{k}": v.numpy() for k, v in torch.load(path).items()}


def main():
    env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE="1")
    out, arrays = {}, {}
    # ---- OpenDPDv2 on APA_200MHz -------------------------------------------------------------------------------------
    C = ["--dataset_name", "APA_200MHz", "--accelerator", "cpu", "--frame_length", "200", "--seed", "0", "--n_epochs", "1",
         "--PA_backbone", "dgru", "--PA_hidden_size", "23"]
    D = ["--DPD_backbone", "deltagru_tcnskip", "--DPD_hidden_size", "15", "--thx", "0.01", "--thh", "0.05", "--batch_size", "64", "--lr", "5e-3"]
    Q = ["--quant", "--n_bits_w", "16", "--n_bits_a", "16", "--quant_dir_label", "w16a16"]
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "_runner.py"), "w").write(RUNNER)
        run = lambda args: subprocess.check_call(["python"] + args, cwd=tmp, env=env, stdout=subprocess.DEVNULL)
        run([os.path.join(REF, "main.py"), "--step", "train_pa", "--batch_size", "256"] + C)
        run([os.path.join(REF, "main.py"), "--step", "train_dpd"] + C + D)
        pre = glob.glob(f"{tmp}/save/APA_200MHz/train_dpd/*/DPD_S_0_M_DELTAGRU_TCNSKIP_H_15_F_200*.pt")[0]
        fhist = glob.glob(f"{tmp}/log/APA_200MHz/train_dpd/*/history/*.csv")[0]
        out["float_stage"] = {"hist": pd.read_csv(fhist).to_dict(orient="list"), "dpd_model": os.path.relpath(pre, tmp), "cmd": " ".join(C + D)}
        run(["_runner.py", "--step", "train_dpd"] + C + D + Q + ["--pretrained_model", pre])
        qpt = [p for p in glob.glob(f"{tmp}/save/APA_200MHz/train_dpd/**/*.pt", recursive=True) if "w16a16" in p][0]
        qhist = [p for p in glob.glob(f"{tmp}/log/APA_200MHz/train_dpd/**/history/*.csv", recursive=True) if "w16a16" in p][0]
        out["qat_stage"] = {"hist": pd.read_csv(qhist).to_dict(orient="list"), "hist_path": os.path.relpath(qhist, tmp),
                            "dpd_model": os.path.relpath(qpt, tmp), "cmd": " ".join(C + D + Q) + " --pretrained_model <float checkpoint>"}
        run(["_runner.py", "--step", "run_dpd"] + C + D + Q)
        csv = [p for p in glob.glob(f"{tmp}/dpd_out/**/*.csv", recursive=True)][0]
        df = pd.read_csv(csv)
        out["run_dpd"] = {"path": os.path.relpath(csv, tmp), "columns": list(df.columns), "rows": int(len(df))}
        arrays["run_dpd_head"] = df.to_numpy()[:4096]
        arrays.update(sd_np(glob.glob(f"{tmp}/save/APA_200MHz/train_pa/*.pt")[0], "pa"))
        arrays.update(sd_np(pre, "fdpd"))
        arrays.update(sd_np(qpt, "qdpd"))
    # ---- a quantised gru DPD on DPA_200MHz (the GRU-swap half of the surgery) ----------------------------------------
    C2 = ["--dataset_name", "DPA_200MHz", "--accelerator", "cpu", "--frame_length", "50", "--seed", "0", "--n_epochs", "1", "--batch_size", "64",
          "--PA_backbone", "gru", "--PA_hidden_size", "11", "--lr", "1e-3"]
    D2 = ["--DPD_backbone", "gru", "--DPD_hidden_size", "11", "--quant", "--n_bits_w", "8", "--n_bits_a", "8", "--quant_dir_label", "w8a8"]
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "_runner.py"), "w").write(RUNNER)
        run = lambda args: subprocess.check_call(["python"] + args, cwd=tmp, env=env, stdout=subprocess.DEVNULL)
        run([os.path.join(REF, "main.py"), "--step", "train_pa"] + C2)
        run(["_runner.py", "--step", "train_dpd"] + C2 + D2)
        qpt = glob.glob(f"{tmp}/save/DPA_200MHz/train_dpd/**/*.pt", recursive=True)[0]
        qhist = glob.glob(f"{tmp}/log/DPA_200MHz/train_dpd/**/history/*.csv", recursive=True)[0]
        out["gru_w8a8_dpa"] = {"hist": pd.read_csv(qhist).to_dict(orient="list"), "hist_path": os.path.relpath(qhist, tmp),
                               "dpd_model": os.path.relpath(qpt, tmp), "cmd": " ".join(C2 + D2)}
        arrays.update(sd_np(glob.glob(f"{tmp}/save/DPA_200MHz/train_pa/*.pt")[0], "pa_dpa"))
        arrays.update(sd_np(qpt, "qgru_dpa"))
    json.dump(out, open(os.path.join(OUT, "ref_runs_v2.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(OUT, "ref_runs_v2.npz"), **arrays)
    for k in ("float_stage", "qat_stage", "gru_w8a8_dpa"):
        print(k, {c: out[k]["hist"].get(c) for c in ("TRAIN_LOSS", "VAL_NMSE", "VAL_EVM", "VAL_ACLR_AVG", "TEST_ACLR_AVG", "N_PARAM", "SP_T_DX", "SP_T_DH")})


if __name__ == "__main__":
    main()
