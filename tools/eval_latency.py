#!/usr/bin/env python3
"""Latency of one evaluation pass (net_eval / run_dpd shapes, train_funcs.py:57-90) per recurrent backbone: torch.no_grad() (the
gate-parallel evaluation kernels where a family has one) beside the same call with gradients enabled (the same kernels writing BPTT
checkpoints on the way; the delta backbones asked for dL/dx take their S16 kernels).  usage: PYTHONPATH=. python tools/eval_latency.py"""
import time

import torch

from opendpd_amd import CoreModel

for bb, H in (("gru", 11), ("dgru", 13), ("dgru", 23), ("qgru", 10), ("qgru_amp1", 16), ("lstm", 14), ("lstm", 24), ("vdlstm", 13), ("vdlstm", 24),
              ("pgjanet", 11), ("deltagru", 15), ("deltagru_tcnskip", 15), ("deltajanet", 15), ("bojanet", 12), ("apnrru", 8), ("dvrjanet", 12), ("mcldnn", 8)):
    for B, T in ((1, 19662), (3, 2560)):
        torch.manual_seed(0)
        net = CoreModel(2, H, 1, bb, **({"num_dvr_units": 3} if bb == "dvrjanet" else {})).cuda().eval()
        x = torch.randn(B, T, 2).cuda() * 0.3
        xg = x.clone().requires_grad_(True)
        best = {}
        for grad in (False, True):
            ts = []
            for _ in range(8):
                torch.cuda.synchronize()
                t = time.perf_counter()
                if grad:
                    net(xg)
                else:
                    with torch.no_grad():
                        net(x)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t)
            best[grad] = min(ts) * 1e3
        print(f"{bb:17s} H{H:<3d} ({B}, {T:5d}, 2): no_grad {best[False]:6.2f} ms   with checkpoints {best[True]:6.2f} ms   ({best[True] / best[False]:.2f}x)", flush=True)
