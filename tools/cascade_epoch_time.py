#!/usr/bin/env python3
"""Wall-clock of a train_dpd epoch's training loop (net_train, frames of resident streams) at the reference's batch size: the native
cascade epoch loop (odpd_train_epoch_cascade: one launch per step body, issued from C++) against the Python-driven chained launches it
replaces (odpd_set_tuning("cascade_one_launch", 0): DPD forward, frozen-PA forward + loss + dL/du, DPD backward on the one-sequence-per-wave
kernels, reduce, clip + AdamW).
usage (GPU box): PYTHONPATH=. python tools/cascade_epoch_time.py"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from opendpd_amd import CascadedModel, CoreModel, _lib
from opendpd_amd.project import DeviceFrameLoader
from opendpd_amd.train_funcs import FusedAdamW, net_train

lib = _lib.load()
dev = torch.device("cuda")
rng = np.random.RandomState(0)
n_s = 23040 + 49           # ~ the train split of DPA_200MHz: 23 040 frames of 50 samples, stride 1
amp, ph = 0.05 + 0.85 * rng.rand(n_s), 2 * np.pi * rng.rand(n_s)
x = np.stack([amp * np.cos(ph), amp * np.sin(ph)], -1)
y = 0.7 * x + 0.05 * rng.randn(n_s, 2)
print("| cascade | frames x length, batch | chained launches from Python (ms / epoch) | one-launch steps, native loop | per step (ms) |")
print("|---|---|---|---|---|")
for dpd_kw, pa_kw in ((dict(hidden_size=15, backbone_type="gru"), dict(hidden_size=23, backbone_type="gru")),
                      (dict(hidden_size=13, backbone_type="dgru"), dict(hidden_size=13, backbone_type="dgru")),
                      (dict(hidden_size=8, backbone_type="dgru"), dict(hidden_size=8, backbone_type="dgru")),
                      (dict(hidden_size=15, backbone_type="deltagru_tcnskip", thx=0.01, thh=0.05), dict(hidden_size=23, backbone_type="dgru"))):
    for T, B in ((50, 64), (200, 64), (200, 256)):
        times = []
        for knob in (0, 1):
            lib.odpd_set_tuning(b"cascade_one_launch", knob)
            torch.manual_seed(0)
            net = CascadedModel(dpd_model=CoreModel(2, num_layers=1, **dpd_kw), pa_model=CoreModel(2, num_layers=1, **pa_kw))
            net.freeze_pa_model()
            net = net.cuda()
            opt = FusedAdamW(net, lr=1e-4)
            loader = DeviceFrameLoader(x, y, T, 1, B, dev, shuffle=True)
            log = {}
            net_train(log, net, loader, opt, torch.nn.MSELoss(), 200.0, dev)       # warm-up epoch
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            net_train(log, net, loader, opt, torch.nn.MSELoss(), 200.0, dev)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) * 1e3)
        lib.odpd_set_tuning(b"cascade_one_launch", 1)
        steps = (loader.n + B - 1) // B
        name = f"{dpd_kw['backbone_type']} {dpd_kw['hidden_size']} -> frozen {pa_kw['backbone_type']} {pa_kw['hidden_size']}"
        print(f"| {name} | {loader.n} x {T}, {B} | {times[0]:.1f} | {times[1]:.1f} | {times[0] / steps:.3f} -> {times[1] / steps:.3f} |", flush=True)
