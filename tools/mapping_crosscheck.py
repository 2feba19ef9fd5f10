#!/usr/bin/env python3
"""Full-size cross-check of the two kernel mappings: at a ragged large batch (default 20 001 x 200, above every S16 crossover) the
default selection (S16 / S16N kernels) and the row-rotated kernels (S16 disabled through the tuning knob) must produce the same
outputs, parameter gradients and dL/dx.  usage: PYTHONPATH=. python tools/mapping_crosscheck.py [B] [T]"""
import sys
import warnings

import numpy as np
import torch

from opendpd_amd import CoreModel, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20001
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
amp = 0.05 + 0.85 * torch.rand(B, T, 1, device="cuda", generator=g)
ph = 2 * np.pi * torch.rand(B, T, 1, device="cuda", generator=g)
x = torch.cat((amp * torch.cos(ph), amp * torch.sin(ph)), -1)
dy = torch.randn(B, T, 2, device="cuda", generator=g) / (B * T)
rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for bb, H, kw in (("gru", 11, {}), ("dgru", 13, {}), ("dgru", 23, {}), ("qgru", 10, {}), ("lstm", 14, {}), ("vdlstm", 13, {}),
                  ("deltagru", 15, dict(thx=0.0, thh=0.0)), ("deltagru_tcnskip", 15, dict(thx=0.0, thh=0.0)), ("pgjanet", 11, {})):
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = CoreModel(2, H, 1, bb, **kw).cuda()
    out = []
    for min_batch in (-1, 1 << 40):
        lib.odpd_set_tuning(b"s16_min_batch", min_batch)
        need_dx = "delta" not in bb
        xt = x.clone().requires_grad_(need_dx)
        for p in net.parameters():
            p.grad = None
        y = net(xt)
        y.backward(dy)
        out.append((y.detach(), torch.cat([p.grad.reshape(-1) for p in net.parameters()]), xt.grad if need_dx else None))
    (y0, g0, d0), (y1, g1, d1) = out
    print(f"{bb:18s} H{H:<3d} S16 vs row-rotated at {B} x {T}:  y {rel(y0, y1):.2e}  grad {rel(g0, g1):.2e}  "
          f"dx {rel(d0, d1) if d0 is not None else 0.0:.2e}", flush=True)
lib.odpd_set_tuning(b"s16_min_batch", -1)

# ---- single-launch train kernels (S16 / S16N / LSTM-S16) against the row-rotated fused or split chain: loss and gradient ----
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step  # noqa: E402

tgt = 0.7 * x + 0.05 * torch.randn(B, T, 2, device="cuda", generator=g)
for bb, H in (("gru", 11), ("dgru", 13), ("dgru", 23), ("qgru", 10), ("qgru_amp1", 16), ("lstm", 14), ("vdlstm", 13), ("lstm", 23)):
    out = []
    for min_batch in (-1, 1 << 40):
        lib.odpd_set_tuning(b"s16_min_batch", min_batch)
        torch.manual_seed(0)
        net = CoreModel(2, H, 1, bb).cuda()
        opt = FusedAdamW(net, lr=0.0)                       # lr 0: parameters stay, opt.grad keeps the reduced gradient
        loss = fused_train_step(opt, x, tgt, "l2", 0.0)
        out.append((float(loss), opt.grad[:opt.backbone.n_flat].clone()))
    (l0, g0), (l1, g1) = out
    print(f"{bb:18s} H{H:<3d} train step, S16 vs row-rotated at {B} x {T}:  loss {abs(l0 - l1) / abs(l1):.2e}  grad {rel(g0, g1):.2e}", flush=True)
lib.odpd_set_tuning(b"s16_min_batch", -1)
