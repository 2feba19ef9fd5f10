#!/usr/bin/env python3
"""A / B of the latency-bound shapes under alternative builds of the library ($OPENDPD_HIP_LIB, one child process per build): the evaluation pass
of a (1, 19 662, 2) segment and the fused train step at 256 x 200, per backbone.   python tools/ab_latency.py [lib.so ...]  ("" = in-tree)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time, json, torch
sys.path.insert(0, %r)
import bench
from opendpd_amd import CoreModel
from opendpd_amd.train_funcs import FusedAdamW, FrameBatch, fused_train_step
dev = torch.device("cuda:0")
out = {}
for bb, H in [a.split(":") for a in sys.argv[1:]]:
    H = int(H)
    torch.manual_seed(0)
    net = CoreModel(2, H, 1, bb).to(dev).eval()
    x = torch.randn(1, 19662, 2, device=dev) * 0.3
    ts = []
    with torch.no_grad():
        for _ in range(12):
            torch.cuda.synchronize(); t = time.perf_counter(); y = net(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    ev = min(ts) * 1e3
    net.train()
    opt = FusedAdamW(net, lr=5e-4)
    framed = bb in ("gru", "dgru", "qgru", "qgru_amp1")
    if framed:
        xs, ys = bench.synth_frames(256, 200, 0, dev, materialize=False)
        xb, tb = FrameBatch(xs, ys, torch.arange(256, device=dev), 200, 1), None
    else:
        xb, tb = bench.synth_frames(256, 200, 0, dev)
    for _ in range(20): fused_train_step(opt, xb, tb, "l2", 200.0, 256 * 200 * 2)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(200): loss = fused_train_step(opt, xb, tb, "l2", 200.0, 256 * 200 * 2)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 200 * 1e3)
    out[f"{bb}{H}"] = {"eval_ms": round(ev, 4), "step_ms": round(best, 5), "y": float(y.double().sum()), "loss": float(loss)}
print(json.dumps(out))
""" % ROOT


def main():
    libs = sys.argv[1:] or [""]
    models = os.environ.get("AB_MODELS", "dgru:13 gru:11 dgru:23").split()
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["OPENDPD_HIP_LIB"] = os.path.abspath(lib)
        else:
            env.pop("OPENDPD_HIP_LIB", None)
        out = subprocess.run([sys.executable, "-c", CHILD, *models], env=env, capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-800:]
        print(f"{os.path.basename(lib) or 'in-tree'}: {line}", flush=True)


if __name__ == "__main__":
    main()
