#!/usr/bin/env python3
"""Fused train step (fwd + loss + BPTT + clip + AdamW) at the reference's own batch sizes: the gate-parallel kernel (gru_gp_train_kernel,
one sequence per wave; lstm_gp_train_kernel for lstm / vdlstm) beside the row-rotated kernels (four sequences per wave: fused for the GRU
family, forward / loss / backward launches for the LSTM family; the delta backbones: split chain either way, their one-sequence-per-wave forward and backward beside the row-rotated ones; odpd_set_tuning gp_max_batch = 0).
(bojanet: boj_gp_train_kernel beside the split S16 chain)
usage: PYTHONPATH=. python tools/gp_train_bench.py [backbone:hidden ...]"""
import ctypes as C
import time

import torch

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from opendpd_amd import CoreModel, _lib
from opendpd_amd.train_funcs import FrameBatch, FusedAdamW, fused_train_step

dev = torch.device("cuda:0")
lib = _lib.load()


def step_ms(bb, H, B, T, gp_max_batch):
    lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(gp_max_batch))
    framed = bb not in ("lstm", "vdlstm", "pgjanet", "deltagru", "deltagru_tcnskip", "deltajanet", "bojanet", "apnrru", "dvrjanet", "mcldnn")            # the LSTM family takes materialised (B, T, 2) frames
    xs, ys = bench.synth_frames(B, T, 0, dev, materialize=not framed)
    torch.manual_seed(0)
    net = CoreModel(2, H, 1, bb, **({"num_dvr_units": 3} if bb == "dvrjanet" else {})).to(dev)
    opt = FusedAdamW(net, lr=5e-4)
    fb, ys = (FrameBatch(xs, ys, torch.arange(B, device=dev), T, 1), None) if framed else (xs, ys)
    for _ in range(5):
        loss = fused_train_step(opt, fb, ys, "l2", 200.0)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            loss = fused_train_step(opt, fb, ys, "l2", 200.0)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / 20 * 1e3)
    return best, float(loss)


CASES = (("gru", 11), ("dgru", 13), ("dgru", 23), ("gru", 23), ("qgru", 10), ("qgru_amp1", 16), ("lstm", 14), ("vdlstm", 13), ("pgjanet", 11), ("deltagru", 15), ("deltagru_tcnskip", 15),
         ("deltajanet", 15), ("bojanet", 12), ("apnrru", 8), ("dvrjanet", 12), ("mcldnn", 8))
if len(sys.argv) > 1:       # e.g. bojanet:12
    CASES = tuple((s.split(":")[0], int(s.split(":")[1])) for s in sys.argv[1:])
for bb, H in CASES:
    for B, T in ((64, 50), (256, 50), (1024, 50), (64, 200), (256, 200), (512, 200), (768, 200), (1024, 200), (2048, 200)):
        a, la = step_ms(bb, H, B, T, 0)
        g, lg = step_ms(bb, H, B, T, 1 << 30)
        d, _ = step_ms(bb, H, B, T, -1)
        print(f"{bb:10s} H{H:<3d} {B:5d} x {T:3d}: row-rotated {a:7.3f} ms   gate-parallel {g:7.3f} ms ({a / g:.2f}x)   default dispatch {d:7.3f} ms   "
              f"loss after 105 steps {la:.6f} | {lg:.6f}", flush=True)
lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(-1))
