#!/usr/bin/env python3
"""Randomised sweep of the QAT QGRU cell (quant/ of the reference) against the C oracle: every hidden size 1..16, both feature
maps, weight / activation widths from {4, 6, 8} (LUT gates: outputs must be BIT-IDENTICAL) and 16 (double-evaluated gates: one
output LSB), random (B, T); outputs, parameter gradients and dL/dx.  usage: PYTHONPATH=. python tools/qat_sweep.py [cases-per-size]"""
import sys
from types import SimpleNamespace

import numpy as np
import torch

from opendpd_amd import CoreModel
from opendpd_amd.quant import get_quant_model
from oracle.oracle import Oracle, make_model

n_per = int(sys.argv[1]) if len(sys.argv) > 1 else 2
o = Oracle("f32")
rng = np.random.RandomState(3)
bad, exact, total, worst = [], 0, 0, [0.0, 0.0, 0.0]
for bb in ("qgru", "qgru_amp1"):
    for H in range(1, 17):
        for case in range(n_per):
            bw, ba = int(rng.choice([4, 6, 8, 16])), int(rng.choice([4, 6, 8, 16]))
            B = int(rng.choice([1, 3, 5, 16, 33])); T = int(rng.choice([1, 4, 5, 31, 33, 64, 130]))
            torch.manual_seed(int(rng.randint(1 << 30)))
            proj = SimpleNamespace(quant=True, n_bits_w=bw, n_bits_a=ba, pretrained_model="")
            q = get_quant_model(proj, CoreModel(2, H, 1, bb)).cuda()
            amp, ph = 0.05 + 0.85 * rng.rand(B, T, 1), 2 * np.pi * rng.rand(B, T, 1)
            x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
            dy = rng.randn(B, T, 2).astype(np.float32)
            q.train()
            xt = torch.from_numpy(x).cuda().requires_grad_(True)
            y = q(xt)
            y.backward(torch.from_numpy(dy).cuda())
            m = make_model(bb, H, bits_w=bw, bits_a=ba)
            p = np.concatenate([v.detach().cpu().numpy().reshape(-1) for v in q.parameters()])
            yo = o.qat_forward(m, p, x)
            go, dxo = o.qat_backward(m, p, x, dy, need_dx=True)
            g = np.concatenate([(v.grad if v.grad is not None else torch.zeros_like(v)).cpu().numpy().reshape(-1) for v in q.parameters()])
            yg = y.detach().cpu().numpy()
            same = bool(np.array_equal(yg, yo))
            rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
            ey, eg, ex = rel(yg, yo), rel(g, go), rel(xt.grad.cpu().numpy(), dxo)
            total += 1; exact += int(same)
            lut = ba <= 8
            if lut and bw <= 8:
                ok = same and eg < 2e-5 and ex < 2e-5
            else:
                ok = ey < 1e-3 and eg < 5e-3 and ex < 5e-3      # wide grids: fp32 accumulation order can move a value across a grid boundary, the recurrence carries it on
                worst = [max(worst[0], ey), max(worst[1], eg), max(worst[2], ex)]
            if not ok:
                bad.append((bb, H, bw, ba, B, T, f"bit-identical {same} y {ey:.2e} g {eg:.2e} dx {ex:.2e}"))
print(f"{total} cases, outputs bit-identical in {exact}; wide-grid cases: worst rel err y {worst[0]:.2e} grad {worst[1]:.2e} dx {worst[2]:.2e}")
print(f"{len(bad)} case(s) beyond tolerance")
for b in bad[:40]:
    print("  ", b)
