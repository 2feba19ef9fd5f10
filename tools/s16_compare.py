#!/usr/bin/env python3
"""Times the two fused GRU-family train kernels (row-rotated 4-seq/wave vs S16 16-seq/wave) over batch sizes.
usage (GPU box): python tools/s16_compare.py [backbone] [hidden]"""
import sys

import torch

from opendpd_amd import CoreModel, _lib
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step

bb = sys.argv[1] if len(sys.argv) > 1 else "dgru"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 13
T = 200
lib = _lib.load()


def run(B, min_batch, occ, iters=5):
    lib.odpd_set_tuning(b"s16_min_batch", min_batch)
    lib.odpd_set_tuning(b"s16_occupancy", occ)
    torch.manual_seed(0)
    net = CoreModel(2, H, 1, bb).cuda()
    opt = FusedAdamW(net, lr=1e-4)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(B, T, 2, device="cuda", generator=g) * 0.8 + 0.05
    t = torch.rand(B, T, 2, device="cuda", generator=g)
    for _ in range(2):
        loss = fused_train_step(opt, x, t, "l2", 200.0)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        loss = fused_train_step(opt, x, t, "l2", 200.0)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters, float(loss)


print(f"{bb} H={H} T={T}: ms per fused step (kernel + reduce + adamw)")
for B in (256, 1024, 4096, 8192, 16384, 32768, 65536):
    a, la = run(B, 1 << 40, 1)
    b, lb = run(B, 0, 1)
    c, lc = run(B, 0, 2)
    print(f"B={B:6d}  rowrot {a:8.3f}  s16/occ1 {b:8.3f}  s16/occ2 {c:8.3f}   Msamples/s {B*T/a/1e3:9.1f} {B*T/b/1e3:9.1f} {B*T/c/1e3:9.1f}"
          f"   loss {la:.6f} {lb:.6f} {lc:.6f}")
