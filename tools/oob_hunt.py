#!/usr/bin/env python3
"""Out-of-bounds hunt (GPU box): every backbone forward + backward on random shapes with the input x and the output gradient dy
placed at the very END of their own 2 MiB device allocations (caching allocator off), and x once more at the very START: a
kernel that reads past either side of a tensor hits an unmapped page and dies with a memory access fault.
An inference call (torch.no_grad(): the evaluation kernels where the shape selects them), the fused train step (x and target placed), the cascade step (random DPD in front of the backbone as frozen PA) and, for
qgru / qgru_amp1, the W8A8 quantisation-aware cell run on the same shapes.
r04: `wide` as the fourth argument draws hidden sizes of the lane-per-unit kernels (33 .. 64; pgjanet 17 .. 32) and, for the backbones that
have them, two stacked layers in every other case; every backbone the surgery has kernels for (gru, dgru, qgru, qgru_amp1, deltagru_tcnskip,
lstm, vdlstm, deltajanet, neuraltx, rvtdcnn, pgjanet) also runs its `--quant` model (forward + backward, inference call, train step).
usage: PYTORCH_NO_CUDA_MEMORY_CACHING=1 PYTHONPATH=. python tools/oob_hunt.py <backbone> <seed> [cases] [wide]"""
import os
import sys
import warnings

import numpy as np
import torch

assert os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING") == "1", "run with PYTORCH_NO_CUDA_MEMORY_CACHING=1"
from opendpd_amd import CascadedModel, CoreModel, _lib  # noqa: E402
from opendpd_amd.quant import get_quant_model  # noqa: E402
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step  # noqa: E402


class _Proj:
    quant = True
    pretrained_model = ""
    n_bits_w = n_bits_a = 8


lib = _lib.load()
bb, seed = sys.argv[1], int(sys.argv[2])
cases = int(sys.argv[3]) if len(sys.argv) > 3 else 150
wide = len(sys.argv) > 4 and sys.argv[4] == "wide"
WIDE_BB = ("gru", "dgru", "qgru", "qgru_amp1", "lstm", "vdlstm", "deltagru", "deltagru_tcnskip", "deltajanet", "pgjanet")
TWO_LAYER_BB = ("gru", "dgru", "qgru", "qgru_amp1", "lstm")
QUANT_MAX_H = {"gru": 32, "dgru": 32, "qgru": 32, "qgru_amp1": 32, "deltagru_tcnskip": 32, "lstm": 64, "vdlstm": 32, "deltajanet": 64, "neuraltx": 64,
               "rvtdcnn": 32, "pgjanet": 32}
assert not wide or bb in WIDE_BB, "no lane-per-unit kernels for this backbone"
rng = np.random.RandomState(seed)
SEG = 2 * 1024 * 1024 // 4


def at_end(t):
    base = torch.empty(SEG, device="cuda")
    v = base[SEG - t.numel():].view(t.shape)
    v.copy_(t)
    return v


def at_start(t):
    base = torch.empty(SEG, device="cuda")
    v = base[:t.numel()].view(t.shape)
    v.copy_(t)
    return v


for it in range(cases):
    H = 11 if bb == "gmp" else int(rng.randint(1, (15 if bb == "apnrru" else 17 if bb in ("pgjanet", "dvrjanet", "bojanet", "mcldnn") else 41 if bb in ("tcnn", "neuraltx") else 33)))
    layers = 1
    if wide:
        H = int(rng.randint(17, 33)) if bb == "pgjanet" else int(rng.randint(33, 65))
        if bb in TWO_LAYER_BB and it % 2:
            H, layers = int(rng.randint(1, 33)), 2
    force = bool(rng.randint(2))
    lib.odpd_set_tuning(b"s16_min_batch", 0 if force else -1)
    B = int(rng.choice([1, 2, 3, 5, 16, 17, 33, 70]))
    T = int(rng.choice([1, 2, 3, 4, 5, 7, 31, 32, 33, 50, 64, 65, 200, 256, 257, 300, 321] + ([513, 700, 1500] if bb in ("gmp", "rvtdcnn", "neuraltx") else [])))
    if B * T > 6000:
        T = max(1, 6000 // B)
    if bb in ("vdlstm", "rvtdcnn") and T < 3:
        T = 3
    if bb == "mcldnn" and T < 4:
        T = 4
    if bb in ("bojanet", "apnrru") and T < 15:
        T = 15 + T
    print(it, bb, H, B, T, force, flush=True)
    torch.manual_seed(it)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = CoreModel(2, H, layers, bb, num_dvr_units=1 + it % 8, **({"thx": 0.01, "thh": 0.02} if "delta" in bb else {})).cuda()
    x0 = (torch.rand(B, T, 2) - 0.5) * 1.6
    x0 = x0 + 0.05 * torch.sign(x0)
    dy = at_end(torch.randn(B, T, 2))
    tgt = torch.randn(B, T, 2) * 0.3
    kw = {"thx": 0.01, "thh": 0.02}
    dpd_bb = ["dgru", "deltagru_tcnskip", "gru", "lstm"][it % 4]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        casc = CascadedModel(dpd_model=CoreModel(2, int(rng.randint(2, 25)), 1, dpd_bb, **(kw if "delta" in dpd_bb else {})),
                             pa_model=CoreModel(2, H, layers, bb, num_dvr_units=1 + it % 8, **(kw if "delta" in bb else {})))
    casc.freeze_pa_model()
    casc = casc.cuda()
    opt, copt = FusedAdamW(net, lr=1e-3), FusedAdamW(casc, lr=1e-3)
    qnet = qopt = None
    if bb in QUANT_MAX_H and layers == 1 and H <= QUANT_MAX_H[bb]:
        _Proj.n_bits_w = _Proj.n_bits_a = 8 if it % 3 else 16
        qnet = get_quant_model(_Proj, CoreModel(2, H, 1, bb, **(kw if "delta" in bb else {}))).cuda().train()
        qopt = FusedAdamW(qnet, lr=1e-3)
    for place in (at_end, at_start):
        x = place(x0).requires_grad_(True)
        y = net(x)
        y.backward(dy)
        with torch.no_grad():           # inference call: the evaluation kernels on few long sequences
            net(place(x0))
        fused_train_step(opt, place(x0), place(tgt), "l2", 200.0)
        if T >= 3 or not ({"vdlstm", "rvtdcnn"} & {bb, dpd_bb}):
            fused_train_step(copt, place(x0), place(tgt), "l2", 200.0)
        if qnet is not None:
            qnet.train()
            xq = place(x0).requires_grad_(True)
            qnet(xq).backward(dy)
            fused_train_step(qopt, place(x0), place(tgt), "l2", 200.0)
            qnet.eval()
            with torch.no_grad():
                qnet(place(x0))
        torch.cuda.synchronize()
print("done")
