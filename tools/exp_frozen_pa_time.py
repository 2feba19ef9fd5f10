#!/usr/bin/env python3
"""Times the fused frozen-PA step (odpd_frozen_loss_dx: forward + loss + dL/du in one launch) of a GRU-family PA model with alternative
builds of the library ($OPENDPD_HIP_LIB), one child process per build.   EXP_H=29 EXP_B=32768 python tools/exp_frozen_pa_time.py [lib.so ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, json, time, ctypes as C, torch
sys.path.insert(0, %r)
from opendpd_amd import CoreModel, _lib
lib = _lib.load()
B, T, H, bb = int(sys.argv[1]), 200, int(sys.argv[2]), sys.argv[3]
torch.manual_seed(0)
pa = CoreModel(2, H, 1, bb).cuda().backbone
g = torch.Generator(device="cuda").manual_seed(1)
u = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.2 + 0.05
t = torch.rand(B, T, 2, device="cuda", generator=g) - 0.5
du = torch.empty_like(u)
rows = int(lib.odpd_frozen_loss_rows(C.byref(pa.desc), B, T))
lr = torch.empty(rows, 4, device="cuda")
ws = torch.empty(int(lib.odpd_ckpt_floats(C.byref(pa.desc), B, T)), device="cuda")
def step():
    _lib.check(lib.odpd_frozen_loss_dx(_lib.stream_ptr(), C.byref(pa.desc), 0, B, T, B * T * 2, _lib.ptr(pa.flat_params()), _lib.ptr(u), _lib.ptr(t), _lib.ptr(du),
                                       _lib.ptr(lr), _lib.ptr(ws)), "frozen")
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print(json.dumps({"ms": (time.perf_counter() - t0) / 10 * 1e3, "loss_sum": float(lr[:, 0].sum()), "du_abs": float(du.abs().sum())}))
""" % ROOT

B, H, bb = os.environ.get("EXP_B", "32768"), os.environ.get("EXP_H", "29"), os.environ.get("EXP_BB", "dgru")
for lib in sys.argv[1:] or [""]:
    env = dict(os.environ)
    if lib:
        env["OPENDPD_HIP_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "-c", CHILD, B, H, bb], env=env, capture_output=True, text=True)
    print(f"{bb} H{H} B{B} {os.path.basename(lib) or 'in-tree'}: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-800:]}", flush=True)
