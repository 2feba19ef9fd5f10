#!/usr/bin/env python3
"""Forward-only (net_eval / run_dpd shape) latency: B segments of T samples through each backbone.
usage (GPU box): PYTHONPATH=. python tools/eval_latency.py"""
import torch

from opendpd_amd import CoreModel

for bb, H in (("gru", 11), ("dgru", 13), ("dgru", 23), ("lstm", 14), ("vdlstm", 13), ("deltagru", 15), ("deltagru_tcnskip", 15),
              ("pgjanet", 11), ("tcnn", 35), ("qgru", 10)):
    kw = dict(thx=0.01, thh=0.05) if "delta" in bb else {}
    net = CoreModel(2, H, 1, bb, **kw).cuda().eval()
    for B, T in ((1, 19662), (3, 2560)):
        x = (torch.rand(B, T, 2, device="cuda") - 0.5) * 1.6
        x = x + 0.05 * torch.sign(x)
        with torch.no_grad():
            for _ in range(2):
                y = net(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                y = net(x)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{bb:18s} H{H:<3d} B={B} T={T:6d}: {ms:8.3f} ms  = {1e3 * ms / T:6.3f} us/step   {B * T / ms / 1e3:8.2f} M samples/s")
