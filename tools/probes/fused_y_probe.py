#!/usr/bin/env python3
"""Recover the outputs y a fused train kernel forms internally from its L2 loss alone (targets = unit impulses): a debugging aid for the
one-launch train kernels, whose forward values never leave the chip.  usage: PYTHONPATH=. python tools/probes/fused_y_probe.py backbone hidden B T"""
import sys
import numpy as np
import torch
from opendpd_amd import CoreModel
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step

bb, H, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
torch.manual_seed(0)
net = CoreModel(2, H, 1, bb, **({"num_dvr_units": 3} if bb == "dvrjanet" else {})).cuda()
with torch.no_grad():
    for k, p in net.named_parameters():
        if "bias" in k:
            p.uniform_(-0.3, 0.3)
import os
if os.environ.get("PROBE_ZERO"):            # e.g. PROBE_ZERO=weight_hh: zero every parameter whose name contains the string
    with torch.no_grad():
        for k, p in net.named_parameters():
            if os.environ["PROBE_ZERO"] in k:
                p.zero_()
                print("zeroed", k)
rng = np.random.RandomState(1)
amp, ph = 0.05 + 0.85 * rng.rand(B, T, 1), 2 * np.pi * rng.rand(B, T, 1)
x = torch.from_numpy(np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)).cuda()
with torch.no_grad():
    y_ref = net(x).cpu().numpy().reshape(-1)
opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
assert opt.has_fused(B, T)
N = B * T * 2
l0 = float(fused_train_step(opt, x, torch.zeros_like(x), "l2", 0.0))
y = np.zeros(N)
for k in range(N):
    tgt = torch.zeros(N, device="cuda")
    tgt[k] = 1.0
    lk = float(fused_train_step(opt, x, tgt.view(B, T, 2), "l2", 0.0))
    y[k] = (l0 - lk + 1.0 / N) * N / 2
np.set_printoptions(precision=5, suppress=True, linewidth=200)
print("fused :", y.reshape(B, T, 2)[0].T)
print("split :", y_ref.reshape(B, T, 2)[0].T)
print("max abs diff", np.abs(y - y_ref).max())
