#!/usr/bin/env python3
"""Times the delta S16 forward / backward launches (odpd_backbone_fwd with checkpoints, odpd_backbone_bwd) of TRes-DeltaGRU H15 at
65 536 x 200 with alternative library builds ($OPENDPD_HIP_LIB), one child process per build.   python tools/exp_delta16_time.py [lib.so ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, json, time, ctypes as C, torch
sys.path.insert(0, %r)
from opendpd_amd import CoreModel, _lib
lib = _lib.load()
B, T, H = int(sys.argv[1]), 200, 15
torch.manual_seed(0)
bb = CoreModel(2, H, 1, sys.argv[2], thx=0.01, thh=0.05).cuda().backbone
g = torch.Generator(device="cuda").manual_seed(1)
x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.2 + 0.05
dy = torch.rand(B, T, 2, device="cuda", generator=g) - 0.5
y = torch.empty_like(x)
ck = torch.empty(int(lib.odpd_ckpt_floats(C.byref(bb.desc), B, T)), device="cuda")
rows = int(lib.odpd_partial_rows(C.byref(bb.desc), B, T, 0))
part = torch.empty(rows, bb.n_flat + 4, device="cuda")
st, p = _lib.stream_ptr, _lib.ptr
def fwd(): _lib.check(lib.odpd_backbone_fwd(st(), C.byref(bb.desc), B, T, p(bb.flat_params()), p(x), p(y), p(ck), None), "fwd")
def bwd(): _lib.check(lib.odpd_backbone_bwd(st(), C.byref(bb.desc), B, T, p(bb.flat_params()), p(x), p(dy), p(ck), p(part), None), "bwd")
out = {}
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    out[name + "_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
out["y_abs"] = float(y.abs().sum()); out["g_abs"] = float(part.sum(0)[:bb.n_flat].abs().sum())
print(json.dumps(out))
""" % ROOT

B, bbn = os.environ.get("EXP_B", "65536"), os.environ.get("EXP_BB", "deltagru_tcnskip")
for lib in sys.argv[1:] or [""]:
    env = dict(os.environ)
    if lib:
        env["OPENDPD_HIP_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "-c", CHILD, B, bbn], env=env, capture_output=True, text=True)
    print(f"{bbn} B{B} {os.path.basename(lib) or 'in-tree'}: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-800:]}", flush=True)
