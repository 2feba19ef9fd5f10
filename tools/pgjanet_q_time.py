#!/usr/bin/env python3
"""Train step and forward pass of the quantised pgjanet (csrc/pgjanet_q.hip) beside the float one at the reference's batch (256 x 200), under
alternative builds of the library ($OPENDPD_HIP_LIB, one child per build).   python tools/pgjanet_q_time.py [lib.so ...]   ("" = in-tree)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time, json, torch
sys.path.insert(0, %r)
import bench
from opendpd_amd import CoreModel
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
from tests.test_quant_more_gpu import _fresh
dev = torch.device("cuda:0")
out = {}
x, t = bench.synth_frames(256, 200, 0, dev)
for H in (11, 16, 24, 32):
    for kind in ("float", "W8A8"):
        torch.manual_seed(0)
        net = (CoreModel(2, H, 1, "pgjanet") if kind == "float" else _fresh("pgjanet", H, 8)).to(dev)
        opt = FusedAdamW(net, lr=5e-4)
        for _ in range(5): fused_train_step(opt, x, t, "l2", 200.0, 256 * 200 * 2)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): loss = fused_train_step(opt, x, t, "l2", 200.0, 256 * 200 * 2)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
        net.eval(); fw = 1e9
        with torch.no_grad():
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter(); y = net(x); torch.cuda.synchronize(); fw = min(fw, (time.perf_counter() - t0) * 1e3)
        out[f"H{H} {kind}"] = {"step_ms": round(best, 4), "fwd_ms": round(fw, 4), "loss": float(loss), "y": float(y.double().sum())}
print(json.dumps(out))
""" % ROOT
for lib in sys.argv[1:] or [""]:
    env = dict(os.environ)
    if lib:
        env["OPENDPD_HIP_LIB"] = os.path.abspath(lib)
    else:
        env.pop("OPENDPD_HIP_LIB", None)
    o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"{os.path.basename(lib) or 'in-tree'}: {o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-800:]}", flush=True)
