#!/usr/bin/env python3
"""N fused train steps of one model in THIS process (for rocprofv3: `rocprofv3 ... -- python3 tools/train_loop.py dgru 23 32768 20`).
usage: tools/train_loop.py <backbone> <hidden> <batch> [steps]      prints one JSON line {ms_per_step, kernel_ms_mean, loss}"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from opendpd_amd import CoreModel  # noqa: E402
from opendpd_amd.train_funcs import FrameBatch, FusedAdamW  # noqa: E402

bb, H, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
NS = int(sys.argv[4]) if len(sys.argv) > 4 else 10
T = 200
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = CoreModel(2, H, 1, bb).to(dev)
opt = FusedAdamW(net, lr=5e-4)
if bb in ("gru", "dgru", "qgru", "qgru_amp1"):
    xs, ys = bench.synth_frames(B, T, 0, dev, materialize=False)
    x, t = FrameBatch(xs, ys, torch.arange(B, device=dev), T, 1), None
else:
    x, t = bench.synth_frames(B, T, 0, dev)
dt, kern_ms, loss = bench.run_steps(opt, x, t, NS, 3, B * T * 2, None, events=True)
print(json.dumps({"backbone": bb, "hidden": H, "batch": B, "steps": NS, "ms_per_step": dt / NS * 1e3, "kernel_ms_mean": kern_ms, "loss": loss}))
