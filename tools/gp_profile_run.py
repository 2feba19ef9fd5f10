#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: 200 fused train steps at the reference's batch shape (256 x 200) per backbone, so the
per-kernel average durations of the gate-parallel train kernels can be read beside the reduce and optimiser launches.
usage: PYTHONPATH=. python tools/gp_profile_run.py [backbone:hidden ...]"""
import torch

import bench
from opendpd_amd import CoreModel
from opendpd_amd.train_funcs import FrameBatch, FusedAdamW, fused_train_step

dev = torch.device("cuda:0")
B, T = 256, 200
import sys
CASES = (("gru", 11), ("dgru", 13), ("lstm", 14), ("vdlstm", 13))
if len(sys.argv) > 1:       # e.g. bojanet:12 apnrru:8 dvrjanet:12 mcldnn:8
    CASES = tuple((s.split(":")[0], int(s.split(":")[1])) for s in sys.argv[1:])
for bb, H in CASES:
    framed = bb in ("gru", "dgru")
    xs, ys = bench.synth_frames(B, T, 0, dev, materialize=not framed)
    torch.manual_seed(0)
    net = CoreModel(2, H, 1, bb, **({"num_dvr_units": 3} if bb == "dvrjanet" else {})).to(dev)
    opt = FusedAdamW(net, lr=5e-4)
    fb, tg = (FrameBatch(xs, ys, torch.arange(B, device=dev), T, 1), None) if framed else (xs, ys)
    for _ in range(200):
        fused_train_step(opt, fb, tg, "l2", 200.0)
    torch.cuda.synchronize()
