#!/usr/bin/env python3
"""HIP-event time of every launch group of the train_dpd cascade step at the reference's batch sizes (latency regime).
usage (GPU box): PYTHONPATH=. python tools/cascade_spans.py"""
import sys

import torch

sys.path.insert(0, ".")
import bench
from types import SimpleNamespace

from opendpd_amd import CascadedModel, CoreModel
from opendpd_amd.quant import get_quant_model
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step

T = 200
for name, dpd_kw, pa_kw, B in (("config 3: TRes-DeltaGRU15 -> frozen DGRU23", dict(hidden_size=15, backbone_type="deltagru_tcnskip", thx=0.01, thh=0.05), dict(hidden_size=23, backbone_type="dgru"), 64),
                               ("default: GRU15 -> frozen GRU23", dict(hidden_size=15, backbone_type="gru"), dict(hidden_size=23, backbone_type="gru"), 256),
                               ("DGRU13 -> frozen DGRU13", dict(hidden_size=13, backbone_type="dgru"), dict(hidden_size=13, backbone_type="dgru"), 256),
                               ("quant_qgru_dpd_regr.sh's float stage: QGRU20 -> frozen DGRU8", dict(hidden_size=20, backbone_type="qgru"), dict(hidden_size=8, backbone_type="dgru"), 64),
                               ("the same, chained launches", dict(hidden_size=20, backbone_type="qgru"), dict(hidden_size=8, backbone_type="dgru"), -64),
                               ("train_all_dpd.sh: LSTM9 -> frozen DGRU8", dict(hidden_size=9, backbone_type="lstm"), dict(hidden_size=8, backbone_type="dgru"), 64),
                               ("the same, chained launches", dict(hidden_size=9, backbone_type="lstm"), dict(hidden_size=8, backbone_type="dgru"), -64),
                               ("config 5: quantisation-aware QGRU10 W8A8 -> frozen DGRU23", dict(hidden_size=10, backbone_type="qgru", bits=8), dict(hidden_size=23, backbone_type="dgru"), 64),
                               ("the same, chained launches", dict(hidden_size=10, backbone_type="qgru", bits=8), dict(hidden_size=23, backbone_type="dgru"), -64),
                               ("OpenDPDv2 QAT stage: quantisation-aware TRes-DeltaGRU15 W16A16 -> frozen DGRU23",
                                dict(hidden_size=15, backbone_type="deltagru_tcnskip", thx=0.01, thh=0.05, bits=16), dict(hidden_size=23, backbone_type="dgru"), 64),
                               ("the same, chained launches", dict(hidden_size=15, backbone_type="deltagru_tcnskip", thx=0.01, thh=0.05, bits=16),
                                dict(hidden_size=23, backbone_type="dgru"), -64)):
    from opendpd_amd import _lib
    _lib.load().odpd_set_tuning(b"cascade_one_launch", 0 if B < 0 else 1)
    B = abs(B)
    torch.manual_seed(0)
    dm = CoreModel(2, num_layers=1, **{k: v for k, v in dpd_kw.items() if k != "bits"})
    if "bits" in dpd_kw:
        dm = get_quant_model(SimpleNamespace(quant=True, n_bits_w=dpd_kw["bits"], n_bits_a=dpd_kw["bits"], pretrained_model=""), dm)
    casc = CascadedModel(dpd_model=dm, pa_model=CoreModel(2, num_layers=1, **pa_kw))
    casc.freeze_pa_model()
    casc = casc.cuda()
    casc.train()
    opt = FusedAdamW(casc, lr=1e-4)
    x, _ = bench.synth_frames(B, T, seed=1, device=torch.device("cuda"))
    t = x.clone()
    for _ in range(5):
        fused_train_step(opt, x, t, "l2", 200.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fused_train_step(opt, x, t, "l2", 200.0)
    e1.record()
    torch.cuda.synchronize()
    spans = bench.cascade_spans(opt, x, t, B * T * 2, n=20)
    print(f"{name}, {B} x {T}: {e0.elapsed_time(e1) / 50:.3f} ms per step; launch groups (ms): " + ", ".join(f"{k} {v:.3f}" for k, v in spans.items()), flush=True)
