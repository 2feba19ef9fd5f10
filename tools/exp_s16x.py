#!/usr/bin/env python3
"""Frozen-PA loss step (odpd_frozen_loss_dx) on the bf16x3 matrix-pipe kernel (gru_s16x.hip) against the exact-fp32 kernel (gru_s16n.hip)
and the fp64 oracle: error table on ragged shapes, then the two kernels timed at EXP_B x 200.   python tools/exp_s16x.py [--no-oracle]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opendpd_amd import CoreModel, _lib  # noqa: E402

lib = _lib.load()


def run(pa, u, t, loss, s16x):
    assert lib.odpd_set_tuning(b"s16x", s16x) == 0
    B, T = u.shape[:2]
    rows = int(lib.odpd_frozen_loss_rows(C.byref(pa.desc), B, T))
    lr = torch.zeros(rows, 4, device="cuda")
    ws = torch.empty(int(lib.odpd_ckpt_floats(C.byref(pa.desc), B, T)), device="cuda")
    du = torch.zeros_like(u)
    _lib.check(lib.odpd_frozen_loss_dx(_lib.stream_ptr(), C.byref(pa.desc), _lib.LOSS_IDS[loss], B, T, B * T * 2, _lib.ptr(pa.flat_params()),
                                       _lib.ptr(u), _lib.ptr(t), _lib.ptr(du), _lib.ptr(lr), _lib.ptr(ws)), "frozen")
    torch.cuda.synchronize()
    return float(lr[:, 0].double().sum()) / (B * T * 2), du


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


lib.odpd_set_tuning(b"s16_min_batch", 0)
use_oracle = "--no-oracle" not in sys.argv
if use_oracle:
    from oracle.oracle import Oracle, make_model
    o64 = Oracle("f64")
print("bb H B T loss | loss: x-vs-n, (x, n)-vs-f64 | du: x-vs-n, x-vs-f64, n-vs-f64")
for bb, H in [("dgru", 23), ("gru", 23), ("dgru", 17), ("dgru", 24), ("qgru", 20), ("qgru_amp1", 21), ("gru", 19)]:
    for B, T, loss in [(37, 70, "l2"), (16 * 9 + 5, 21, "l2"), (33, 201, "l1"), (64, 200, "l2")]:
        torch.manual_seed(H * 7 + B)
        pa = CoreModel(2, H, 1, bb).cuda().backbone
        rng = np.random.RandomState(H + B)
        u = (rng.uniform(0.1, 0.8, (B, T, 2)) * rng.choice([-1.0, 1.0], (B, T, 2))).astype(np.float32)
        t = (0.5 * rng.randn(B, T, 2)).astype(np.float32)
        ug, tg = torch.from_numpy(u).cuda(), torch.from_numpy(t).cuda()
        lx, dx = run(pa, ug, tg, loss, 1)
        ln, dn = run(pa, ug, tg, loss, 0)
        line = f"{bb} {H} {B} {T} {loss} | {abs(lx - ln) / ln:.2e}"
        if use_oracle:
            mp = make_model(bb, H)
            pp = pa.flat_params().detach().cpu().numpy().astype(np.float64)
            y, _ = o64.forward(mp, pp, u.astype(np.float64))
            lo, dy = o64.loss(loss, y, t.astype(np.float64))
            _, du = o64.backward(mp, pp, u.astype(np.float64), dy)
            du = torch.from_numpy(np.asarray(du)).cuda()
            line += f" ({abs(lx - lo) / lo:.2e}, {abs(ln - lo) / lo:.2e}) | {rel(dx, dn):.2e} {rel(dx.double(), du):.2e} {rel(dn.double(), du):.2e}"
        else:
            line += f" | {rel(dx, dn):.2e}"
        print(line, flush=True)

B, T = int(os.environ.get("EXP_B", "65536")), 200
for bb, H in [("dgru", 23), ("gru", 23)]:
    torch.manual_seed(0)
    pa = CoreModel(2, H, 1, bb).cuda().backbone
    g = torch.Generator(device="cuda").manual_seed(1)
    u = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.2 + 0.05
    t = torch.rand(B, T, 2, device="cuda", generator=g) - 0.5
    du = torch.empty_like(u)
    rows = int(lib.odpd_frozen_loss_rows(C.byref(pa.desc), B, T))
    lr = torch.empty(rows, 4, device="cuda")
    for s16x in (1, 0, 1, 0):
        lib.odpd_set_tuning(b"s16x", s16x)
        ws = torch.empty(int(lib.odpd_ckpt_floats(C.byref(pa.desc), B, T)), device="cuda")

        def step():
            _lib.check(lib.odpd_frozen_loss_dx(_lib.stream_ptr(), C.byref(pa.desc), 0, B, T, B * T * 2, _lib.ptr(pa.flat_params()), _lib.ptr(u), _lib.ptr(t),
                                               _lib.ptr(du), _lib.ptr(lr), _lib.ptr(ws)), "frozen")
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f"{bb} H{H} {B}x{T} s16x={s16x}: {ms:.3f} ms  loss_sum {float(lr[:, 0].sum()):.6f} du_abs {float(du.abs().sum()):.4f}", flush=True)
