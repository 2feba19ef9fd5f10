#!/usr/bin/env python3
"""train_dpd step (DPD -> frozen PA, five launches) throughput at a large batch.
usage (GPU box): PYTHONPATH=. python tools/cascade_timing.py [B]"""
import sys

import torch

from opendpd_amd import CascadedModel, CoreModel, _lib
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = 200
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.2
x = x + 0.05 * torch.sign(x)
t = x * 1.0
for dpd, dh, pa, ph in (("dgru", 13, "dgru", 13), ("dgru", 13, "gru", 11), ("dgru", 13, "dgru", 23), ("deltagru_tcnskip", 15, "dgru", 13)):
    for mb, tag in ((-1, "default (S16 where H <= 16)"), (1 << 40, "row-rotated only")):
        lib.odpd_set_tuning(b"s16_min_batch", mb)
        torch.manual_seed(0)
        net = CascadedModel(dpd_model=CoreModel(2, dh, 1, dpd, thx=0.01, thh=0.05), pa_model=CoreModel(2, ph, 1, pa))
        net.freeze_pa_model()
        net = net.cuda()
        opt = FusedAdamW(net, lr=1e-4)
        for _ in range(2):
            fused_train_step(opt, x, t, "l2", 200.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            loss = fused_train_step(opt, x, t, "l2", 200.0)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{dpd} H{dh} -> frozen {pa} H{ph}, B={B}: {ms:7.3f} ms/step = {B * T / ms / 1e6:6.2f} G samples/s   [{tag}]  loss {float(loss):.5f}")
lib.odpd_set_tuning(b"s16_min_batch", -1)
