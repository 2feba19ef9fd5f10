#!/usr/bin/env python3
"""GMP kernel timings (HIP events, median): forward, split backward (weight gradient; with dL/dx), fused train kernel.
usage (GPU box): PYTHONPATH=. python tools/gmp_bench.py [B] [T]"""
import ctypes as C
import sys

import torch

from opendpd_amd import CoreModel, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lib = _lib.load()
torch.manual_seed(0)
net = CoreModel(2, 11, 1, "gmp").cuda()
bb = net.backbone
g = torch.Generator(device="cuda").manual_seed(1)
x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.6
t = torch.randn(B, T, 2, device="cuda", generator=g) * 0.3
y, dx = torch.empty_like(x), torch.empty_like(x)
rows = int(lib.odpd_partial_rows(C.byref(bb.desc), B, T, 0))
part = torch.empty(rows, 499, device="cuda")
flat = bb.flat_params()
st = _lib.stream_ptr()
P = _lib.ptr


def timeit(fn, n=15):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


d = bb.desc
res = {
    "fwd": timeit(lambda: lib.odpd_backbone_fwd(st, C.byref(d), B, T, P(flat), P(x), P(y), None, None)),
    "wgrad": timeit(lambda: lib.odpd_backbone_bwd(st, C.byref(d), B, T, P(flat), P(x), P(t), None, P(part), None)),
    "dx": timeit(lambda: lib.odpd_backbone_bwd(st, C.byref(d), B, T, P(flat), P(x), P(t), None, None, P(dx))),
    "fused": timeit(lambda: lib.odpd_train_fwd_bwd(st, C.byref(d), 0, B, T, B * T * 2, P(flat), P(x), P(t), P(part), None)),
}
n = B * T
print(f"gmp B={B} T={T} rows={rows}: " + "  ".join(f"{k} {v * 1e3:.1f} us ({n / v / 1e6:.1f} G samples/s)" for k, v in res.items()))
