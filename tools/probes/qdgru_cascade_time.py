#!/usr/bin/env python3
"""Step time of a train_dpd step with a quantised dgru DPD in front of a frozen dgru PA: one launch (qat_cascade_kernel<K_DGRU, ..>) against the
chained launches.  usage (GPU box): PYTHONPATH=. python tools/probes/qdgru_cascade_time.py"""
import ctypes as C
from types import SimpleNamespace

import torch

from opendpd_amd import CascadedModel, CoreModel, _lib
from opendpd_amd.quant import get_quant_model
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step

lib = _lib.load()
for H, Hp, bits, B, T in ((13, 23, 8, 64, 200), (13, 23, 16, 64, 200), (13, 13, 8, 64, 50), (10, 23, 8, 256, 200)):
    for one in (1, 0):
        lib.odpd_set_tuning(b"cascade_one_launch", C.c_int64(one))
        torch.manual_seed(0)
        dpd = get_quant_model(SimpleNamespace(quant=True, n_bits_w=bits, n_bits_a=bits, pretrained_model=""), CoreModel(2, H, 1, "dgru"))
        net = CascadedModel(dpd_model=dpd, pa_model=CoreModel(2, Hp, 1, "dgru"))
        net.freeze_pa_model()
        net = net.cuda().train()
        opt = FusedAdamW(net, lr=1e-3)
        x = (torch.rand(B, T, 2, device="cuda") - 0.5) * 1.2
        t = (torch.rand(B, T, 2, device="cuda") - 0.5)
        for _ in range(5):
            fused_train_step(opt, x, t, "l2", 200.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fused_train_step(opt, x, t, "l2", 200.0)
        e1.record()
        torch.cuda.synchronize()
        print(f"quantised dgru H{H} W{bits}A{bits} -> frozen dgru H{Hp}, {B} x {T}: {'one launch' if one else 'chained   '} {e0.elapsed_time(e1) / 50:.3f} ms per step", flush=True)
lib.odpd_set_tuning(b"cascade_one_launch", C.c_int64(1))
