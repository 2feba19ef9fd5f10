#!/usr/bin/env python3
"""Times the SPLIT entry points (odpd_backbone_fwd with checkpoints, odpd_backbone_bwd from dL/dy with parameter gradients) of one model at a
large batch through the raw C ABI, with the bf16x3 family of hidden 17 .. 24 on and off (knob "s16x_train").
usage: tools/split_time.py [backbone] [hidden] [batch]"""
import ctypes as C
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from opendpd_amd import CoreModel, _lib  # noqa: E402

bb, H, B = (sys.argv[1] if len(sys.argv) > 1 else "dgru"), int(sys.argv[2]) if len(sys.argv) > 2 else 23, int(sys.argv[3]) if len(sys.argv) > 3 else 32768
T = 200
lib = _lib.load()
torch.manual_seed(0)
net = CoreModel(2, H, 1, bb).cuda().backbone
g = torch.Generator(device="cuda").manual_seed(1)
x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.2 + 0.05
dy = torch.randn(B, T, 2, device="cuda", generator=g) * 1e-3
y = torch.empty_like(x)
P = net.n_flat
for knob in (1, 0, 1, 0):
    assert lib.odpd_set_tuning(b"s16x_train", knob) == 0
    ck = torch.empty(int(lib.odpd_ckpt_floats(C.byref(net.desc), B, T)), device="cuda")
    rows = int(lib.odpd_partial_rows(C.byref(net.desc), B, T, 0))
    part = torch.empty(rows, P + _lib.LOSS_COLS, device="cuda")

    def fwd():
        _lib.check(lib.odpd_backbone_fwd(_lib.stream_ptr(), C.byref(net.desc), B, T, _lib.ptr(net.flat_params()), _lib.ptr(x), _lib.ptr(y), _lib.ptr(ck), None), "fwd")

    def bwd():
        _lib.check(lib.odpd_backbone_bwd(_lib.stream_ptr(), C.byref(net.desc), B, T, _lib.ptr(net.flat_params()), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(ck),
                                         _lib.ptr(part), None), "bwd")
    out = {}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        out[name + "_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    out.update(knob=knob, y_abs=float(y.abs().sum()), grad_abs=float(part.double().sum(0)[:P].abs().sum()))
    print(json.dumps(out), flush=True)
