#!/usr/bin/env python3
"""Latency of one evaluation pass (net_eval / run_dpd shapes) of the quantisation-aware models: gru / qgru / qgru_amp1 / deltagru_tcnskip of
<= 16 units on the one-sequence-per-wave engines (csrc/gru_cascade.hip: qat_eval_kernel), the quantised dgru on its 16-sequences-per-wave kernels.
usage (GPU box): PYTHONPATH=. python tools/qat_eval_latency.py"""
import time, torch
from types import SimpleNamespace
from opendpd_amd import CoreModel
from opendpd_amd.quant import get_quant_model
for bb, H, bits, kw in (("qgru", 10, 8, {}), ("qgru", 10, 16, {}), ("gru", 11, 8, {}), ("dgru", 13, 8, {}), ("deltagru_tcnskip", 15, 16, dict(thx=0.01, thh=0.05)), ("deltagru_tcnskip", 15, 8, dict(thx=0.01, thh=0.05))):
    torch.manual_seed(0)
    net = get_quant_model(SimpleNamespace(quant=True, n_bits_w=bits, n_bits_a=bits, pretrained_model=""), CoreModel(2, H, 1, bb, **kw)).cuda().eval()
    for B, T in ((1, 19662), (3, 2560)):
        x = torch.randn(B, T, 2).cuda() * 0.3
        ts = []
        for _ in range(6):
            torch.cuda.synchronize(); t = time.perf_counter()
            with torch.no_grad():
                net(x)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        print(f"{bb:18s} H{H} W{bits}A{bits} ({B}, {T}): eval {min(ts)*1e3:.2f} ms", flush=True)
