#!/usr/bin/env python3
"""Train-step and forward throughput of every HIP backbone (one process, HIP events) -> markdown table.
usage (GPU box): PYTHONPATH=. python tools/family_table.py [--big 32768] [--out file.md]"""
import argparse

import torch

from opendpd_amd import CascadedModel, CoreModel
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step

ap = argparse.ArgumentParser()
ap.add_argument("--big", type=int, default=32768)
ap.add_argument("--T", type=int, default=200)
ap.add_argument("--out", default=None)
ap.add_argument("--only", default=None, help="comma-separated backbone names (skips the cascade row)")
a = ap.parse_args()
T = a.T


def timeit(fn, n, w=2):
    """median over n calls, each bracketed by its own pair of HIP events (robust against one-off allocator / clock hiccups)"""
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


CASES = [("gru", 11, {}), ("dgru", 13, {}), ("dgru", 23, {}), ("qgru", 10, {}), ("lstm", 14, {}), ("vdlstm", 13, {}),
         ("deltagru", 15, dict(thx=0.01, thh=0.05)), ("deltagru_tcnskip", 15, dict(thx=0.01, thh=0.05)), ("pgjanet", 11, {}),
         ("tcnn", 35, {}), ("gmp", 11, {}), ("rvtdcnn", 25, {}), ("rvtdcnn", 6, {}), ("neuraltx", 36, {}), ("deltajanet", 15, {}), ("dvrjanet", 12, dict(num_dvr_units=3)), ("bojanet", 12, {}), ("apnrru", 8, {}), ("mcldnn", 8, {}), ("qgru W8A8 (QAT)", 10, dict(qat=(8, 8))), ("qgru_amp1 W8A8 (QAT)", 10, dict(qat=(8, 8)))]
rows = []
only = a.only.split(",") if a.only else None
for bb, H, kw in CASES:
    if only and bb.split()[0] not in only:
        continue
    torch.manual_seed(0)
    qat = kw.pop("qat", None)
    net = CoreModel(2, H, 1, bb.split()[0], **kw)
    if qat:
        from types import SimpleNamespace
        from opendpd_amd.quant import get_quant_model
        net = get_quant_model(SimpleNamespace(quant=True, n_bits_w=qat[0], n_bits_a=qat[1], pretrained_model=""), net)
    net = net.cuda()
    P = sum(p.numel() for p in net.parameters())
    opt = FusedAdamW(net, lr=1e-4)
    cells = []
    for B in (256, a.big):
        x, t = data(B)
        ms = timeit(lambda: fused_train_step(opt, x, t, "l2", 200.0), 21 if B <= 1024 else 7)
        with torch.no_grad():
            msf = timeit(lambda: net(x), 21 if B <= 1024 else 7)
        cells += [ms, B * T / ms / 1e3, msf, B * T / msf / 1e3]
    f256, fbig = opt.has_fused(256, T), opt.has_fused(a.big, T)
    kind = ("fused (1 launch)" if f256 and fbig else "split (fwd, loss, bwd)" if not (f256 or fbig)
            else "fused at 256, split at the large batch" if f256 else "split at 256, fused at the large batch")
    rows.append((f"{bb} H{H}", P, kind, *cells))
# train_dpd cascade of BASELINE config 3
torch.manual_seed(0)
if only:
    for r in rows:
        print(r)
    raise SystemExit(0)
casc = CascadedModel(dpd_model=CoreModel(2, 15, 1, "deltagru_tcnskip", thx=0.01, thh=0.05), pa_model=CoreModel(2, 23, 1, "dgru"))
casc.freeze_pa_model()
casc = casc.cuda()
opt = FusedAdamW(casc, lr=1e-4)
cells = []
for B in (64, a.big):
    x, t = data(B)
    ms = timeit(lambda: fused_train_step(opt, x, t, "l2", 200.0), 21 if B <= 1024 else 7)
    with torch.no_grad():
        msf = timeit(lambda: casc(x), 21 if B <= 1024 else 7)
    cells += [ms, B * T / ms / 1e3, msf, B * T / msf / 1e3]
rows.append(("train_dpd: TRes-DeltaGRU15 -> frozen DGRU23 (B = 64 | big)", 999, "cascade (DPD fwd, frozen-PA fwd+loss+dL/du, DPD bwd)", *cells))

hdr = (f"| backbone | params | train step | B=256: step ms | M samples/s | fwd ms | M samples/s | B={a.big}: step ms | M samples/s | fwd ms "
       f"| M samples/s |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
body = "".join(f"| {r[0]} | {r[1]} | {r[2]} | " + " | ".join(f"{v:.3f}" if i % 2 == 0 else f"{v:.0f}" for i, v in enumerate(r[3:])) + " |\n"
               for r in rows)
txt = (f"# Train-step (fwd + MSE + BPTT + clip 200 + AdamW) and forward-only throughput, T = {T}, fp32, 1 x MI355X\n\n"
       "`tools/family_table.py`, HIP events around repeated calls, inputs resident in HBM.\n\n" + hdr + body)
print(txt)
if a.out:
    open(a.out, "w").write(txt)
