#!/usr/bin/env python3
"""BASELINE config 3 train_dpd step (TRes-DeltaGRU15 -> frozen DGRU23) for profiling.
usage (GPU box): PYTHONPATH=. python3 tools/cascade_config3.py [B] [steps]"""
import sys

import torch

from opendpd_amd import CascadedModel, CoreModel
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
T = 200
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.2
x = x + 0.05 * torch.sign(x)
t = x * 1.0
torch.manual_seed(0)
net = CascadedModel(dpd_model=CoreModel(2, 15, 1, "deltagru_tcnskip", thx=0.01, thh=0.05), pa_model=CoreModel(2, 23, 1, "dgru"))
net.freeze_pa_model()
net = net.cuda()
opt = FusedAdamW(net, lr=1e-4)
for _ in range(2):
    fused_train_step(opt, x, t, "l2", 200.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    loss = fused_train_step(opt, x, t, "l2", 200.0)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
print(f"config 3 cascade, B={B}: {ms:.3f} ms/step = {B * T / ms / 1e6:.2f} G samples/s, loss {float(loss):.5f}")
