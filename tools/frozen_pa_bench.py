#!/usr/bin/env python3
"""odpd_frozen_loss_dx (frozen PA of train_dpd: forward + loss + dL/du) at the reference's batch sizes: the one-sequence-per-wave
gate-parallel kernel against the row-rotated kernel it replaces there (odpd_set_tuning("gp_max_batch", 0) switches it off).
usage (GPU box): PYTHONPATH=. python tools/frozen_pa_bench.py"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from opendpd_amd import CoreModel, _lib

lib = _lib.load()


def time_it(desc, flat, B, T, iters=200):
    rows = int(lib.odpd_frozen_loss_rows(C.byref(desc), B, T))
    u = torch.rand(B, T, 2, device="cuda") - 0.5
    t = torch.rand(B, T, 2, device="cuda") - 0.5
    du = torch.empty_like(u)
    lr = torch.empty(rows, _lib.LOSS_COLS, device="cuda")
    ws = torch.empty(max(int(lib.odpd_train_workspace_floats(C.byref(desc), B, T)), 1), device="cuda")

    def run():
        rc = lib.odpd_frozen_loss_dx(_lib.stream_ptr(), C.byref(desc), 0, B, T, B * T * 2, _lib.ptr(flat), _lib.ptr(u), _lib.ptr(t), _lib.ptr(du),
                                     _lib.ptr(lr), _lib.ptr(ws))
        assert rc == 0, rc
    for _ in range(10):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, du.clone()


print("| PA | batch x frame | row-rotated ms | gate-parallel ms | ratio |")
print("|---|---|---|---|---|")
for bb, H in (("gru", 11), ("dgru", 13), ("qgru", 10), ("gru", 23), ("dgru", 23), ("gru", 32), ("dgru", 32)):
    pa = CoreModel(2, H, 1, bb).cuda()
    flat = pa.backbone.flat_params()
    for B, T in ((64, 200), (256, 200), (512, 200), (1024, 200), (64, 50), (1024, 50)):
        lib.odpd_set_tuning(b"gp_max_batch", 0)
        t_rot, d_rot = time_it(pa.backbone.desc, flat, B, T)
        lib.odpd_set_tuning(b"gp_max_batch", 1 << 20)      # (a set value also lifts the hidden 17..32 exclusion of the plain GRU)
        t_gp, d_gp = time_it(pa.backbone.desc, flat, B, T)
        lib.odpd_set_tuning(b"gp_max_batch", -1)
        print(f"| {bb} {H} | {B} x {T} | {t_rot:.3f} | {t_gp:.3f} | {t_rot / t_gp:.2f} |", flush=True)
