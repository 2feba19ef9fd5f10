#!/usr/bin/env python3
"""Removal experiments on BASELINE config 4's kernel (lstm16_train_kernel<VDLSTM, 1 tile, K-packed>, train_pa VDLSTM H13 at 32 768 x 200):
builds the library once per removed component (-D flags, own object directories and output paths: the in-tree library is not touched) and
times each against the in-tree build with tools/exp_time.py.  Results of the variant builds are wrong by construction; timing only.
   python tools/exp_cfg4_removal.py build      (here, no GPU)          python tools/exp_cfg4_removal.py time      (GPU box)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {"no_mfma": ["-DODPD_X_NOMFMA"], "no_operand_stream": ["-DODPD_X_L16_NOSTREAM"], "no_transcendentals": ["-DODPD_X_NOTRANS"],
            "no_checkpoint_io": ["-DODPD_X_L16_NOCKPT"]}
OUTDIR = os.path.join(ROOT, "build", "exp_cfg4")

if sys.argv[1:] == ["build"]:
    from opendpd_amd import build as hb
    os.makedirs(OUTDIR, exist_ok=True)
    for name, flags in VARIANTS.items():
        print(name, hb.build(extra_flags=tuple(flags), out=os.path.join(OUTDIR, f"lib_{name}.so")), flush=True)
else:
    env = dict(os.environ, EXP_B="32768", EXP_H="13", EXP_BB="vdlstm", EXP_STEPS="20")
    libs = [""] + [os.path.join(OUTDIR, f"lib_{n}.so") for n in VARIANTS] + [""]
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "exp_time.py")] + [l for l in libs if l] , env=env)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "exp_time.py")], env=env)
