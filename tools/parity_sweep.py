#!/usr/bin/env python3
"""Randomised parity sweep of every HIP backbone against the C oracle (GPU box): all hidden sizes of the kernels' envelope,
random batch / frame length, the three kernel mappings (default dispatch — the one-sequence-per-wave kernels at these shapes —, S16 forced, row-rotated forced).  Prints the worst relative errors per backbone
and every case beyond tolerance.  usage: PYTHONPATH=. python tools/parity_sweep.py [cases-per-size] [mid]
`mid`: batches of 300 / 1000 / 4803 sequences x 3..19 steps instead — tens to hundreds of 16-sequence groups, i.e. several workgroups
and partial rows per launch, the range between the ragged small shapes and the full-size runs of tests/test_fullsize_gpu.py"""
import sys
import warnings

import numpy as np
import torch

from opendpd_amd import CoreModel, _lib
from oracle.oracle import Oracle, make_model

n_per = int(sys.argv[1]) if len(sys.argv) > 1 else 2
MID = len(sys.argv) > 2 and sys.argv[2] == "mid"
lib = _lib.load()
o = Oracle("f32")
WIDE = len(sys.argv) > 2 and sys.argv[2] == "wide"      # `wide`: only the sizes the lane-per-unit kernels serve (33 .. 64 units: csrc/*_wide.hip)
SIZES = {"gru": range(1, 65), "dgru": range(1, 65), "qgru": range(1, 65), "qgru_amp1": range(1, 65), "lstm": range(1, 65),
         "vdlstm": range(1, 65), "deltagru": range(1, 65), "deltagru_tcnskip": range(1, 65), "pgjanet": range(1, 33),
         "tcnn": list(range(1, 40)) + [48, 63, 64], "gmp": [11] * 12, "rvtdcnn": range(1, 33), "deltajanet": range(1, 65),
         "neuraltx": list(range(1, 40)) + [48, 63, 64], "dvrjanet": range(1, 17), "bojanet": range(1, 17), "apnrru": range(1, 15), "mcldnn": range(1, 17)}
rng = np.random.RandomState(0)
bad, kinks, illcond, derailed, worst = [], [], [], [], {}
if WIDE:
    SIZES = {k: range(33, 65) for k in ("gru", "dgru", "qgru", "qgru_amp1", "lstm", "vdlstm", "deltagru", "deltagru_tcnskip", "deltajanet")}
    SIZES["pgjanet"] = range(17, 33)
for bb, sizes in SIZES.items():
    for H in sizes:
        for case in range(n_per):
            for force in (False, True, "row-rotated"):      # default dispatch (one-sequence-per-wave kernels at these shapes) | S16 forced | four-sequence waves
                lib.odpd_set_tuning(b"s16_min_batch", 0 if force is True else -1)
                lib.odpd_set_tuning(b"gp_max_batch", 0 if force == "row-rotated" else -1)
                B = int(rng.choice([1, 2, 3, 5, 16, 17, 33, 70]))
                T = int(rng.choice([1, 2, 3, 4, 5, 7, 31, 32, 33, 50, 64, 65, 200, 257, 300]))
                if MID:
                    B, T = int(rng.choice([300, 1000, 4803])), int(rng.choice([3, 4, 5, 8, 16, 19]))
                elif B * T > 6000:
                    T = max(1, 6000 // B)
                if bb in ("vdlstm", "rvtdcnn") and T < 3:
                    T = 3       # the 3-sample circular pad needs T >= 3 (vdlstm.py:66-74); shorter frames are refused (EINVAL)
                if bb == "mcldnn" and T < 4:
                    T = 4       # the circular window takes the frame's own last four samples (mcldnn.py:115-118)
                if bb in ("bojanet", "apnrru") and T < 15:
                    T = 15 + T  # the reference cuts its 15-sample zero pad from the frame itself (bojanet.py:72-73)
                kw = dict(thx=float(rng.choice([0.0, 0.01, 0.05])), thh=float(rng.choice([0.0, 0.02, 0.1]))) if "delta" in bb else {}
                K = int(rng.randint(1, 9)) if bb == "dvrjanet" else 0
                torch.manual_seed(int(rng.randint(1 << 30)))
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    net = CoreModel(2, H, 1, bb, num_dvr_units=K or None, **kw).cuda()
                with torch.no_grad():
                    for k, p in net.named_parameters():
                        if "bias" in k:
                            p.uniform_(-0.3, 0.3)
                        if k.startswith("backbone.conv_"):       # NeuralTX FIR taps: large enough for every path to count
                            p.uniform_(-0.6, 0.6)
                        if k.startswith("backbone.fir_"):        # BOJANET taps (gain 0.1): let the envelopes reach the gates
                            p.mul_(4.0)
                        if k == "backbone.rru.Z":                # APNRRU: Z = 0 at construction switches the deep cell off
                            p.uniform_(-0.6, 0.6)
                        if k == "backbone.cs":                   # DVR coefficients: bounded gain (standard-normal ones make the recurrence chaotic)
                            p.mul_(min(1.0, 1.5 / float(p.abs().sum())))
                amp, ph = 0.05 + 0.85 * rng.rand(B, T, 1), 2 * np.pi * rng.rand(B, T, 1)
                x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
                dy = rng.randn(B, T, 2).astype(np.float32)
                need_dx = "delta" not in bb or force is True      # delta: dL/dx lives in the S16 kernels (ODPD_FLAG_NEED_DX routes there)
                xt = torch.from_numpy(x).cuda().requires_grad_(need_dx)
                try:
                    y = net(xt)
                    y.backward(torch.from_numpy(dy).cuda())
                except Exception as e:      # noqa: BLE001
                    bad.append((bb, H, B, T, force, kw, f"EXC {e}"))
                    continue
                m = make_model(bb, H, kw.get("thx", 0), kw.get("thh", 0), bits_w=K)
                p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
                yo, so = o.forward(m, p, x)
                go, dxo = o.backward(m, p, x, dy, need_dx=need_dx)
                g = np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in net.parameters()])
                rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
                ey, eg = rel(y.detach().cpu().numpy(), yo), rel(g, go)
                ex = rel(xt.grad.cpu().numpy(), dxo) if need_dx else 0.0
                flips = 0
                if "delta" in bb:
                    st = net.backbone.statistics
                    flips = abs(st["num_dx_zeros"] - so[0]) + abs(st["num_dh_zeros"] - so[2])
                tol_y, tol_g = (2e-5, 3e-4) if flips == 0 else (5e-3, 5e-2)
                w = worst.setdefault(bb, [0.0, 0.0, 0.0])
                if "delta" in bb and flips == 0 and ey >= tol_y:
                    # two flipped threshold decisions can cancel in the counters: a flip derails ONE sequence from that step on
                    per_seq = np.abs(y.detach().cpu().numpy() - yo).reshape(B, -1).max(1) / max(np.abs(yo).max(), 1e-30)
                    if int((per_seq > tol_y).sum()) <= 2:
                        flips, tol_y, tol_g = 2, 5e-2, 5e-2
                if flips == 0:
                    w[0], w[1], w[2] = max(w[0], ey), max(w[1], eg), max(w[2], ex)
                if not (ey < tol_y and eg < tol_g and ex < tol_g) or not np.isfinite([ey, eg, ex]).all() or flips > 4:
                    # how many sequences carry the output error: a threshold decision that flipped (delta backbones; two flips can
                    # cancel in the counters) derails ONE sequence from that step on, a kernel defect hits them all
                    per_seq = np.abs(y.detach().cpu().numpy() - yo).reshape(B, -1).max(1) / max(np.abs(yo).max(), 1e-30)
                    nbad = int((per_seq > tol_y).sum())
                    if "delta" in bb and nbad <= 2 and ey < 5e-2 and eg < 5e-2:
                        continue
                    if "delta" in bb and nbad <= 2 and flips > 4:
                        # ONE (two) sequence(s) off while the others agree to tolerance and the decision counters differ by many: an early
                        # flip whose changed state flipped later decisions too (large thresholds, long frames) — listed, not counted
                        derailed.append((bb, H, B, T, force, kw, f"y {ey:.2e} g {eg:.2e} counters differ by {flips:.0f}, sequences off {nbad}/{B}"))
                        continue
                    # an activation kink (relu of the DGRU head, hardswish, |.| of the DVR, a demodulator output next to 0) within rounding
                    # of its corner makes the GRADIENT of one sequence discontinuous while the outputs agree: the oracle's own fp64
                    # gradient then moves by as much under a 2e-6 relative change of the input
                    if need_dx and ey < tol_y:
                        per_dx = np.abs(xt.grad.cpu().numpy() - dxo).reshape(B, -1).max(1) / max(np.abs(dxo).max(), 1e-30)
                        if int((per_dx > tol_g).sum()) <= 2:
                            o64 = Oracle("f64")
                            f8 = lambda v: np.asarray(v, dtype=np.float64)
                            g8 = [o64.backward(m, f8(p), f8(x) * sc, f8(dy))[1] for sc in (1.0, 1 + 2e-6, 1 - 2e-6)]
                            jump = max(float(np.abs(v - g8[0]).max() / max(np.abs(g8[0]).max(), 1e-30)) for v in g8[1:])
                            if jump > 0.3 * max(ex, eg):
                                kinks.append((bb, H, B, T, force, f"g {eg:.2e} dx {ex:.2e} in {int((per_dx > tol_g).sum())} sequence(s); fp64 oracle jumps {jump:.2e}"))
                                continue
                    # a demodulator output / amplitude next to 0 somewhere in the batch: the fp32 oracle itself is then that far from fp64
                    o64 = Oracle("f64")
                    f8 = lambda v: np.asarray(v, dtype=np.float64)
                    y8, _ = o64.forward(m, f8(p), f8(x))
                    g8, dx8 = o64.backward(m, f8(p), f8(x), f8(dy), need_dx=need_dx)
                    cy, cg, cx = rel(yo, y8), rel(go, g8), (rel(dxo, dx8) if need_dx else 0.0)
                    if flips == 0 and ey < tol_y + 20 * cy and eg < tol_g + 20 * cg and ex < tol_g + 20 * cx:
                        illcond.append((bb, H, B, T, force, f"y {ey:.2e} g {eg:.2e} dx {ex:.2e}; fp32 oracle vs fp64: y {cy:.2e} g {cg:.2e} dx {cx:.2e}"))
                        continue
                    bad.append((bb, H, B, T, force, kw, f"y {ey:.2e} g {eg:.2e} dx {ex:.2e} flips {flips} sequences off {nbad}/{B}"))
    print(f"{bb:18s} worst rel err  y {worst[bb][0]:.2e}  grad {worst[bb][1]:.2e}  dx {worst[bb][2]:.2e}", flush=True)
lib.odpd_set_tuning(b"s16_min_batch", -1)
lib.odpd_set_tuning(b"gp_max_batch", -1)
print(f"{len(kinks)} case(s) on an activation kink (the oracle's own gradient is discontinuous there): {kinks}")
print(f"{len(derailed)} thresholded case(s) in which a flipped delta decision derailed one sequence beyond 5e-2 (all other sequences within tolerance): {derailed}")
print(f"{len(illcond)} ill-conditioned case(s) (the fp32 oracle itself is that far from the fp64 one): {illcond}")
print(f"{len(bad)} case(s) beyond tolerance")
for b in bad[:60]:
    print("  ", b)
