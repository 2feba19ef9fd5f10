#!/usr/bin/env python3
"""Train-step / forward time of the quantisation-aware models (HIP events, inputs resident), per kernel mapping.
usage (GPU box): PYTHONPATH=. python tools/qat_bench.py [--B 32768] [--T 200] [--cases qgru:10:8,deltagru_tcnskip:15:16,...]"""
import argparse
import ctypes as C
from types import SimpleNamespace

import torch

from opendpd_amd import CoreModel, _lib
from opendpd_amd.quant import get_quant_model
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step

ap = argparse.ArgumentParser()
ap.add_argument("--B", default="256,32768")
ap.add_argument("--T", type=int, default=200)
ap.add_argument("--cases", default="qgru:10:8,qgru:10:16,qgru:20:8,gru:11:8,dgru:13:8,dgru:23:8,deltagru_tcnskip:15:8,deltagru_tcnskip:15:16,deltagru_tcnskip:30:8")
ap.add_argument("--mappings", default="auto")
a = ap.parse_args()
T = a.T
lib = _lib.load()


def timeit(fn, n, w=2):
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


def data(B):
    g = torch.Generator(device="cuda").manual_seed(B)
    x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.6
    x = x + 0.05 * torch.sign(x)
    return x, torch.randn(B, T, 2, device="cuda", generator=g) * 0.3


print(f"| model | bits | mapping | B | train step ms | G samples/s | forward ms | G samples/s |\n|---|---|---|---|---|---|---|---|")
for case in a.cases.split(","):
    bb, H, bits = case.split(":")
    H, bits = int(H), int(bits)
    kw = dict(thx=0.01, thh=0.05) if "delta" in bb else {}
    for mapping in a.mappings.split(","):
        lib.odpd_set_tuning(b"s16_min_batch", C.c_int64({"auto": -1, "s16": 0, "rot": 1 << 30}[mapping]))
        torch.manual_seed(0)
        net = get_quant_model(SimpleNamespace(quant=True, n_bits_w=bits, n_bits_a=bits, pretrained_model=""), CoreModel(2, H, 1, bb, **kw)).cuda()
        opt = FusedAdamW(net, lr=1e-4)
        for B in [int(b) for b in a.B.split(",")]:
            x, t = data(B)
            net.train()
            ms = timeit(lambda: fused_train_step(opt, x, t, "l2", 200.0), 15 if B <= 1024 else 5)
            net.eval()
            with torch.no_grad():
                msf = timeit(lambda: net(x), 15 if B <= 1024 else 5)
            print(f"| {bb} H{H} | W{bits}A{bits} | {mapping} | {B} | {ms:.3f} | {B * T / ms / 1e6:.2f} | {msf:.3f} | {B * T / msf / 1e6:.2f} |", flush=True)
lib.odpd_set_tuning(b"s16_min_batch", C.c_int64(-1))
