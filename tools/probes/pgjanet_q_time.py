#!/usr/bin/env python3
"""Train-step time of pgjanet, float (janet_family / janet_wide kernels) and quantised (csrc/pgjanet_q.hip), at the reference's batch.
usage (GPU box): PYTHONPATH=. python tools/probes/pgjanet_q_time.py"""
from types import SimpleNamespace

import torch

from opendpd_amd import CoreModel
from opendpd_amd.quant import get_quant_model
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step

for H in (11, 24, 32):
    for quant in (False, True):
        torch.manual_seed(0)
        net = CoreModel(2, H, 1, "pgjanet")
        if quant:
            net = get_quant_model(SimpleNamespace(quant=True, n_bits_w=8, n_bits_a=8, pretrained_model=""), net)
        net = net.cuda().train()
        opt = FusedAdamW(net, lr=1e-3)
        x, t = torch.rand(256, 200, 2, device="cuda") - 0.5, torch.rand(256, 200, 2, device="cuda") - 0.5
        for _ in range(3):
            fused_train_step(opt, x, t, "l2", 200.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fused_train_step(opt, x, t, "l2", 200.0)
        e1.record()
        torch.cuda.synchronize()
        kind = "W8A8 " if quant else "float"
        print(f"pgjanet H{H} {kind} 256 x 200 train step {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
