#!/usr/bin/env python3
"""train_dpd step of bash_scripts/quant_qgru_dpd_regr.sh's QAT stage shapes (quantised QGRU W16A16 of 6 .. 30 units in front of a frozen DGRU8,
64 frames of 50 samples): the one-launch cascade step against the chained launches.  usage (GPU box): PYTHONPATH=. python tools/qat_regr_shapes.py"""
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
import bench
from opendpd_amd import CascadedModel, CoreModel, _lib
from opendpd_amd.quant import get_quant_model
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step

lib = _lib.load()
for H in (6, 9, 13, 20, 30):
    res = []
    for knob in (1, 0):
        lib.odpd_set_tuning(b"cascade_one_launch", knob)
        torch.manual_seed(3)
        qdpd = get_quant_model(SimpleNamespace(quant=True, n_bits_w=16, n_bits_a=16, pretrained_model=""), CoreModel(2, H, 1, "qgru"))
        casc = CascadedModel(dpd_model=qdpd, pa_model=CoreModel(2, 8, 1, "dgru"))
        casc.freeze_pa_model()
        casc = casc.cuda()
        casc.train()
        opt = FusedAdamW(casc, lr=1e-4)
        x, _ = bench.synth_frames(64, 50, seed=1, device=torch.device("cuda"))
        t = x.clone()
        for _ in range(5):
            fused_train_step(opt, x, t, "l2", 200.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fused_train_step(opt, x, t, "l2", 200.0)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 100)
    lib.odpd_set_tuning(b"cascade_one_launch", 1)
    print(f"quantised QGRU{H} W16A16 -> frozen DGRU8, 64 x 50: chained launches {res[1]:.3f} ms per step, one launch {res[0]:.3f} ms", flush=True)
