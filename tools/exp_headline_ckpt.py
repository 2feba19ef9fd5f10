#!/usr/bin/env python3
"""Removal experiments on the headline kernel's BPTT checkpoint I/O (gru16_train_kernel, DGRU H13, 65 536 x 200): variant libraries without the
checkpoint stores of the forward pass / without the checkpoint loads of the backward pass / without both (timing only: wrong results).
   python tools/exp_headline_ckpt.py build   (here)        python tools/exp_headline_ckpt.py time   (GPU box)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {"no_ckpt_stores": ["-DODPD_X_S16_NOST"], "no_ckpt_loads": ["-DODPD_X_S16_NOLD"], "no_ckpt_io": ["-DODPD_X_S16_NOST", "-DODPD_X_S16_NOLD"]}
OUTDIR = os.path.join(ROOT, "build", "exp_headline")
if sys.argv[1:] == ["build"]:
    from opendpd_amd import build as hb
    os.makedirs(OUTDIR, exist_ok=True)
    for name, flags in VARIANTS.items():
        print(name, hb.build(extra_flags=tuple(flags), out=os.path.join(OUTDIR, f"lib_{name}.so")), flush=True)
else:
    env = dict(os.environ, EXP_B="65536", EXP_H="13", EXP_BB="dgru", EXP_STEPS="20")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "exp_time.py"), ""] + [os.path.join(OUTDIR, f"lib_{n}.so") for n in VARIANTS], env=env)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "exp_time.py")], env=env)
