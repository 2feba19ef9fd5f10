#!/usr/bin/env python3
"""Runs the frozen-PA loss step (odpd_frozen_loss_dx) EXP_N times at EXP_B x 200 for profiling.
EXP_BB / EXP_H: the PA model (default dgru 23); EXP_S16X=0: the exact-fp32 kernel (gru_s16n.hip) instead of the bf16x3 one."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opendpd_amd import CoreModel, _lib  # noqa: E402

lib = _lib.load()
B, T, H, bb = int(os.environ.get("EXP_B", "65536")), 200, int(os.environ.get("EXP_H", "23")), os.environ.get("EXP_BB", "dgru")
lib.odpd_set_tuning(b"s16x", int(os.environ.get("EXP_S16X", "1")))
torch.manual_seed(0)
pa = CoreModel(2, H, 1, bb).cuda().backbone
g = torch.Generator(device="cuda").manual_seed(1)
u = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.2 + 0.05
t = torch.rand(B, T, 2, device="cuda", generator=g) - 0.5
du = torch.empty_like(u)
rows = int(lib.odpd_frozen_loss_rows(C.byref(pa.desc), B, T))
lr = torch.empty(rows, 4, device="cuda")
ws = torch.empty(int(lib.odpd_ckpt_floats(C.byref(pa.desc), B, T)), device="cuda")
for _ in range(int(os.environ.get("EXP_N", "12"))):
    _lib.check(lib.odpd_frozen_loss_dx(_lib.stream_ptr(), C.byref(pa.desc), 0, B, T, B * T * 2, _lib.ptr(pa.flat_params()), _lib.ptr(u), _lib.ptr(t),
                                       _lib.ptr(du), _lib.ptr(lr), _lib.ptr(ws)), "frozen")
torch.cuda.synchronize()
print("loss_sum", float(lr[:, 0].sum()))
