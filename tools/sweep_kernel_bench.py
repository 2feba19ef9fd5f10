#!/usr/bin/env python3
"""GPU time of one lockstep epoch (odpd_train_epoch_sweep: per step one fused train launch for K runs, one reduction, one clip + AdamW) against the
solo epoch (odpd_train_epoch), DGRU H13 at 256 x 200 on synthetic streams: ms per step for K = 1 .. 64.   EXP_BB / EXP_H / EXP_BATCH / EXP_T"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from opendpd_amd import CoreModel, _lib  # noqa: E402
from opendpd_amd.train_funcs import FusedAdamW  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
bb, H = os.environ.get("EXP_BB", "dgru"), int(os.environ.get("EXP_H", "13"))
B, T, steps = int(os.environ.get("EXP_BATCH", "256")), int(os.environ.get("EXP_T", "200")), 40
n = B * steps
xs, ys = bench.synth_frames(n, T, 0, dev, materialize=False)


def make(K):
    runs = []
    for k in range(K):
        torch.manual_seed(k)
        net = CoreModel(2, H, 1, bb).to(dev)
        opt = FusedAdamW(net, lr=5e-4)
        opt._ensure(dev)
        rows = int(lib.odpd_partial_rows(C.byref(net.backbone.desc), B, T, 1))
        rows = max(rows, int(lib.odpd_sweep_partial_rows(C.byref(net.backbone.desc), B, T, 1)))
        runs.append(dict(net=net, opt=opt, part=torch.empty(rows, net.backbone.n_flat + 4, device=dev), losses=torch.empty(steps, device=dev),
                         ws=torch.empty(max(1, int(lib.odpd_sweep_workspace_floats(C.byref(net.backbone.desc), B, T, 1))), device=dev),
                         order=torch.randperm(n, generator=torch.Generator().manual_seed(k)).to(dev)))
    return runs


def epoch_sweep(runs, scratch, flags=0):
    K = len(runs)
    table = (_lib.SweepRun * K)()
    for k, r in enumerate(runs):
        table[k] = _lib.SweepRun(r["net"].backbone.flat_params().data_ptr(), r["opt"].grad.data_ptr(), r["opt"].exp_avg.data_ptr(), r["opt"].exp_avg_sq.data_ptr(),
                                 r["part"].data_ptr(), r["losses"].data_ptr(), None, r["ws"].data_ptr(), r["order"].data_ptr(), 5e-4)
    fr = _lib.Frames(xs.data_ptr(), ys.data_ptr(), None, n, T, 1, 0, 0)
    _lib.check(lib.odpd_train_epoch_sweep(_lib.stream_ptr(), C.byref(runs[0]["net"].backbone.desc), K, table, 0, C.byref(fr), B, 1, 0.9, 0.999, 1e-8, 0.01,
                                          200.0, flags, C.c_void_p(scratch.data_ptr())), "sweep")


def epoch_solo(r):
    fr = _lib.Frames(xs.data_ptr(), ys.data_ptr(), r["order"].data_ptr(), n, T, 1, 0, 0)
    _lib.check(lib.odpd_train_epoch(_lib.stream_ptr(), C.byref(r["net"].backbone.desc), 0, C.byref(fr), B, _lib.ptr(r["net"].backbone.flat_params()),
                                    _lib.ptr(r["opt"].grad), _lib.ptr(r["opt"].exp_avg), _lib.ptr(r["opt"].exp_avg_sq), 1, 5e-4, 0.9, 0.999, 1e-8, 0.01, 200.0,
                                    _lib.ptr(r["part"]), None, _lib.ptr(r["losses"])), "solo")


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


r1 = make(1)
t_solo = timed(lambda: epoch_solo(r1[0]))
print(f"{bb} H{H} {B} x {T}: solo epoch {1e3 * t_solo / steps:.4f} ms per step")
for K in (1, 2, 4, 8, 16, 32, 64):
    runs = make(K)
    scratch = torch.empty(int(lib.odpd_sweep_scratch_bytes(K, steps)), dtype=torch.uint8, device=dev)
    t = timed(lambda: epoch_sweep(runs, scratch))
    t16 = timed(lambda: epoch_sweep(runs, scratch, 1))
    print(f"K = {K:2d}: exact {1e3 * t / steps:.4f} ms per step = {t / t_solo:.2f} x solo ({K * t_solo / t:.1f} x the throughput) | "
          f"S16 {1e3 * t16 / steps:.4f} ms per step = {t16 / t_solo:.2f} x solo ({K * t_solo / t16:.1f} x the throughput)", flush=True)
