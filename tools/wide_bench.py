#!/usr/bin/env python3
"""Train-step time of the lane-per-unit kernels (csrc/gru_wide.hip, lstm_wide.hip: 33 .. 64 hidden units; gru_layers2.hip: two layers) at the reference's batch
shapes, next to the ATen restatement they replaced (backbones/wide.py + torch.optim.AdamW).  usage (GPU box): PYTHONPATH=. python tools/wide_bench.py"""
import warnings

import torch

from opendpd_amd import CoreModel
from opendpd_amd.backbones import wide as W
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step


def timeit(fn, n=10, w=2):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return ts[len(ts) // 2]


print("| model | B x T | kernels: step ms | ATen restatement: step ms |\n|---|---|---|---|")
for bb, H, NL in (("gru", 48, 1), ("dgru", 40, 1), ("dgru", 64, 1), ("qgru", 36, 1), ("lstm", 48, 1), ("vdlstm", 40, 1), ("deltagru", 40, 1), ("deltagru_tcnskip", 48, 1), ("pgjanet", 24, 1), ("gru", 8, 2), ("gru", 23, 2), ("qgru", 32, 2)):
    for B, T in ((64, 50), (256, 200), (2048, 200)):
        g = torch.Generator(device="cuda").manual_seed(B)
        x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.6
        x = x + 0.05 * torch.sign(x)
        t = torch.randn(B, T, 2, device="cuda", generator=g) * 0.3
        torch.manual_seed(0)
        net = CoreModel(2, H, NL, bb, **({"thx": 0.01, "thh": 0.05} if "delta" in bb else {})).cuda()
        opt = FusedAdamW(net, lr=1e-4)
        ms = timeit(lambda: fused_train_step(opt, x, t, "l2", 200.0))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            old = dict(W.KERNEL_HIDDEN_LIMIT)
            W.KERNEL_HIDDEN_LIMIT.update(gru=32, dgru=32, qgru=32, qgru_amp1=32, lstm=32, vdlstm=32, deltagru=32, deltagru_tcnskip=32, pgjanet=16)
            old2, W.TWO_LAYER_KERNELS = W.TWO_LAYER_KERNELS, ()
            ref = CoreModel(2, H, NL, bb, **({"thx": 0.01, "thh": 0.05} if "delta" in bb else {})).cuda()
            W.KERNEL_HIDDEN_LIMIT.update(old)
            W.TWO_LAYER_KERNELS = old2
        topt = torch.optim.AdamW(ref.parameters(), lr=1e-4)

        def aten_step():
            topt.zero_grad()
            loss = torch.nn.functional.mse_loss(ref(x), t)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(ref.parameters(), 200.0)
            topt.step()
        msa = timeit(aten_step, n=5, w=1)
        print(f"| {bb} H{H}{' x 2 layers' if NL == 2 else ''} | {B} x {T} | {ms:.3f} | {msa:.2f} |", flush=True)
