#!/usr/bin/env python3
"""Times the fused DGRU-H13 train step (65 536 x 200, the bench workload) with alternative builds of the library
($OPENDPD_HIP_LIB), one child process per build.  Used for kernel experiments (e.g. -DODPD_EXP builds of csrc/: timing only, their
results are not meaningful).   python tools/exp_time.py [lib.so ...]   (no argument: the in-tree library)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, json, torch
sys.path.insert(0, %r)
import bench
from opendpd_amd import CoreModel
from opendpd_amd.train_funcs import FusedAdamW, FrameBatch
B, T, H = int(sys.argv[1]), 200, int(sys.argv[2])
dev = torch.device("cuda:0")
import os
framed = os.environ.get("EXP_FRAMED", "1" if sys.argv[3] in ("gru", "dgru", "qgru", "qgru_amp1") else "0") == "1"
torch.manual_seed(0)
net = CoreModel(2, H, 1, sys.argv[3]).to(dev)
opt = FusedAdamW(net, lr=5e-4)
if framed:
    xs, ys = bench.synth_frames(B, T, 0, dev, materialize=False)
    x, t = FrameBatch(xs, ys, torch.arange(B, device=dev), T, 1), None
else:          # kernels that take (B, T, 2) frames
    x, t = bench.synth_frames(B, T, 0, dev)
NS = int(os.environ.get("EXP_STEPS", "10"))
dt, kern_ms, loss = bench.run_steps(opt, x, t, NS, 3, B * T * 2, None, events=True)
print(json.dumps({"ms_per_step": dt / NS * 1e3, "kernel_ms_mean": kern_ms, "loss": loss}))
""" % ROOT


def main():
    libs = sys.argv[1:] or [""]
    B, H, bb = os.environ.get("EXP_B", "65536"), os.environ.get("EXP_H", "13"), os.environ.get("EXP_BB", "dgru")
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["OPENDPD_HIP_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "-c", CHILD, B, H, bb], env=env, capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-600:]
        print(f"{os.path.basename(lib) or 'in-tree'}: {line}", flush=True)


if __name__ == "__main__":
    main()
