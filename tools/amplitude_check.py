#!/usr/bin/env python3
"""Accuracy against the float64 oracle when the WHOLE signal is small (input scale 1, 0.1, 0.01): the backbones whose features all
scale with the amplitude (gru, qgru*, lstm, vdlstm, pgjanet) need a tanh with relative accuracy near 0; both kernel mappings.
usage: PYTHONPATH=. python tools/amplitude_check.py"""
import numpy as np
import torch

from opendpd_amd import CoreModel, _lib
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
from oracle.oracle import Oracle, make_model

lib = _lib.load()
B, T = 64, 200
o = Oracle("f64")
for bb, H in (("gru", 11), ("gru", 23), ("dgru", 13), ("qgru", 10), ("qgru_amp1", 20), ("lstm", 14), ("lstm", 23), ("vdlstm", 13), ("pgjanet", 11),
              ("deltagru", 15)):
    for scale in (1.0, 0.1, 0.01):
        g = torch.Generator().manual_seed(0)
        amp, ph = (0.05 + 0.85 * torch.rand(B, T, 1, generator=g)) * scale, 2 * np.pi * torch.rand(B, T, 1, generator=g)
        x = torch.cat((amp * torch.cos(ph), amp * torch.sin(ph)), -1)
        tgt = 0.8 * x
        torch.manual_seed(0)
        net = CoreModel(2, H, 1, bb).cuda()
        p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
        m = make_model(bb, H)
        yo, _ = o.forward(m, p.astype(np.float64), x.numpy().astype(np.float64))
        lo, dyo = o.loss("l2", yo, tgt.numpy().astype(np.float64))
        go, _ = o.backward(m, p.astype(np.float64), x.numpy().astype(np.float64), dyo, need_dx=False)
        res = []
        for mb in (0, 1 << 40):
            lib.odpd_set_tuning(b"s16_min_batch", mb)
            with torch.no_grad():
                y = net(x.cuda()).cpu().numpy()
            opt = FusedAdamW(net, lr=0.0)
            fused_train_step(opt, x.cuda(), tgt.cuda(), "l2", 0.0)
            gr = opt.grad[:opt.backbone.n_flat].cpu().numpy()
            res += [np.abs(y - yo).max() / np.abs(yo).max(), np.abs(gr - go).max() / np.abs(go).max()]
        print(f"{bb:10s} H{H:<3d} input scale {scale:5.2f}:  S16 y {res[0]:.2e} train-step grad {res[1]:.2e}   row-rotated y {res[2]:.2e} grad {res[3]:.2e}",
              flush=True)
lib.odpd_set_tuning(b"s16_min_batch", -1)
