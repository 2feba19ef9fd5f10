#!/usr/bin/env python3
"""Times the individual kernels of the DGRU path at a given batch (HIP events on the current stream)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opendpd_amd import CoreModel
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step


def timeit(fn, n=10, w=2):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--hidden", type=int, default=13)
ap.add_argument("--bb", default="dgru")
ap.add_argument("--T", type=int, default=200)
a = ap.parse_args()
torch.manual_seed(0)
net = CoreModel(2, a.hidden, 1, a.bb).cuda()
x = torch.rand(a.batch, a.T, 2, device="cuda") * 0.8 + 0.05
t = torch.rand(a.batch, a.T, 2, device="cuda")
opt = FusedAdamW(net, lr=1e-4)
n = a.batch * a.T


def fwd_only():
    with torch.no_grad():
        net(x)


def fwd_bwd():
    for p in net.parameters():
        p.grad = None
    y = net(x)
    y.backward(t)


ms_f = timeit(fwd_only)
ms_fb = timeit(fwd_bwd)
ms_fused = timeit(lambda: fused_train_step(opt, x, t, "l2", 200.0))
print(f"{a.bb} H{a.hidden} B{a.batch} T{a.T}: fwd(no ckpt) {ms_f:.3f} ms ({n/ms_f/1e6:.2f} GS/s) | "
      f"fwd+ckpt+bwd unfused {ms_fb:.3f} ms ({n/ms_fb/1e6:.2f} GS/s) | fused step {ms_fused:.3f} ms ({n/ms_fused/1e6:.2f} GS/s)")
