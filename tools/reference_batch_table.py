#!/usr/bin/env python3
"""One table for the latency regime: the train step (fwd + loss + BPTT + reduce + clip + AdamW through fused_train_step, default dispatch) of every
registry backbone at the reference's own batch shapes — 64 / 256 frames of 50 / 200 samples (train_funcs.py:28-48; bash_scripts/*.sh) —, and one
evaluation pass over a (1, 19 662, 2) segment (net_eval).  Markdown on stdout.  usage: PYTHONPATH=. python tools/reference_batch_table.py"""
import time
import warnings

import torch

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from opendpd_amd import CoreModel
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step

warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
CASES = (("gru", 11), ("dgru", 13), ("dgru", 23), ("qgru", 10), ("qgru_amp1", 16), ("lstm", 14), ("vdlstm", 13), ("pgjanet", 11), ("deltagru", 15),
         ("deltagru_tcnskip", 15), ("tcnn", 35), ("gmp", 11), ("rvtdcnn", 6), ("neuraltx", 12), ("deltajanet", 15), ("dvrjanet", 12), ("bojanet", 12),
         ("apnrru", 8), ("mcldnn", 8))
SHAPES = ((64, 50), (256, 50), (64, 200), (256, 200))


def step_ms(net, B, T):
    xs, ys = bench.synth_frames(B, T, 0, dev, materialize=True)
    opt = FusedAdamW(net, lr=5e-4)
    fused = opt.has_fused(B, T)
    for _ in range(5):
        fused_train_step(opt, xs, ys, "l2", 200.0)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            fused_train_step(opt, xs, ys, "l2", 200.0)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / 20 * 1e3)
    return best, fused


print("| backbone | parameters | " + " | ".join(f"{B} x {T} (ms)" for B, T in SHAPES) + " | launches per step body | eval (1, 19 662, 2) (ms) |")
print("|---|---|" + "---|" * (len(SHAPES) + 2))
for bb, H in CASES:
    torch.manual_seed(0)
    net = CoreModel(2, H, 1, bb, **({"num_dvr_units": 3} if bb == "dvrjanet" else {})).to(dev)
    cells, fused = [], None
    for B, T in SHAPES:
        ms, fused = step_ms(net, B, T)
        cells.append(f"{ms:.3f}")
    net.eval()
    x = torch.randn(1, 19662, 2, device=dev) * 0.3
    ts = []
    with torch.no_grad():
        for _ in range(6):
            torch.cuda.synchronize()
            t = time.perf_counter()
            net(x)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
    print(f"| {bb} H{H} | {sum(p.numel() for p in net.parameters())} | " + " | ".join(cells) + f" | {'one' if fused else 'forward, loss, backward'} | {min(ts) * 1e3:.2f} |", flush=True)
