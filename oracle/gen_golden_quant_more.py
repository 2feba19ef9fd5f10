#!/usr/bin/env python3
"""Golden vectors for the GENERAL quantisation surgery (TEST INFRASTRUCTURE — runs only in the build container).

quant/quant_envs.py:114-130 swaps every nn.GRU for the Python GRU of GRUCells, :290-306 swaps every Sigmoid / Tanh / Add / Mul
module and every nn.Linear: `--quant` is therefore accepted on gru and dgru (cell quantised, dgru's fc_hid / fc_out become
INT_Linear), on qgru at any hidden size (bash_scripts/quant_qgru_dpd_regr.sh:74 sweeps 6..30) and on deltagru_tcnskip, whose
layer routes its gate arithmetic through such modules (deltagru_tcnskip.py:156-162, 286-290) — the OpenDPDv2 recipe
(bash_scripts/OpenDPDv2.sh:84-117: W16A16 from a float checkpoint).  This script RUNS the reference for those models and
stores inputs, train- / eval-mode outputs, gradients, three clip + AdamW steps, the sparsity counters (delta) and, for
deltagru_tcnskip, the same from a `--pretrained_model` float checkpoint.  The same is stored for lstm / vdlstm, where the surgery
swaps only the nn.Linear heads (fc_out; fc_lambda_1 / fc_lambda_2 / fc_out) and the nn.LSTM core stays float.

Usage:  python oracle/gen_golden_quant_more.py [name ...]
"""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402  (puts /root/reference on the path, bridges quant.Sqrt / quant.Pow)

import torch  # noqa: E402
import quant  # noqa: E402

CASES = [
    # name, backbone, hidden, bits, thx, thh, pretrained
    ("quant_gru_h11_w8a8", "gru", 11, 8, 0, 0, False),
    ("quant_gru_h23_w8a8", "gru", 23, 8, 0, 0, False),
    ("quant_gru_h11_w16a16", "gru", 11, 16, 0, 0, False),
    ("quant_dgru_h13_w8a8", "dgru", 13, 8, 0, 0, False),
    ("quant_dgru_h23_w8a8", "dgru", 23, 8, 0, 0, False),
    ("quant_dgru_h13_w16a16", "dgru", 13, 16, 0, 0, False),
    ("quant_qgru_h20_w8a8", "qgru", 20, 8, 0, 0, False),
    ("quant_qgru_h30_w8a8", "qgru", 30, 8, 0, 0, False),
    ("quant_qgru_amp1_h20_w16a16", "qgru_amp1", 20, 16, 0, 0, False),
    ("quant_tres_h15_w8a8_th", "deltagru_tcnskip", 15, 8, 0.01, 0.05, False),
    ("quant_tres_h15_w8a8_dense", "deltagru_tcnskip", 15, 8, 0.0, 0.0, False),
    ("quant_tres_h15_w16a16_th", "deltagru_tcnskip", 15, 16, 0.01, 0.05, False),
    ("quant_tres_h30_w8a8_th", "deltagru_tcnskip", 30, 8, 0.005, 0.02, False),
    ("quant_tres_h15_w16a16_pre", "deltagru_tcnskip", 15, 16, 0.01, 0.05, True),
    ("quant_tres_h15_w8a8_pre", "deltagru_tcnskip", 15, 8, 0.01, 0.05, True),
    # backbones in which the surgery finds only nn.Linear layers to swap (quant_envs.py:40-60): float nn.LSTM core, INT_Linear heads
    ("quant_lstm_h14_w8a8", "lstm", 14, 8, 0, 0, False),
    ("quant_lstm_h14_w16a16", "lstm", 14, 16, 0, 0, False),
    ("quant_lstm_h24_w8a8", "lstm", 24, 8, 0, 0, False),
    ("quant_lstm_h40_w8a8", "lstm", 40, 8, 0, 0, False),
    ("quant_vdlstm_h13_w8a8", "vdlstm", 13, 8, 0, 0, False),
    ("quant_vdlstm_h13_w16a16", "vdlstm", 13, 16, 0, 0, False),
    ("quant_deltajanet_h12_w8a8", "deltajanet", 12, 8, 0, 0, False),       # custom float cell (nn.Parameter gates), INT_Linear fc_out
    ("quant_deltajanet_h40_w16a16", "deltajanet", 40, 16, 0, 0, False),
    # neuraltx: the surgery's layer map holds nn.Conv2d and nn.Linear (quant_envs.py:145-148) — the Conv1d stack stays float, IQ_match -> INT_Linear
    ("quant_neuraltx_h12_w8a8", "neuraltx", 12, 8, 0, 0, False),
    ("quant_neuraltx_h20_w16a16", "neuraltx", 20, 16, 0, 0, False),
    # rvtdcnn (feed-forward): Conv2d -> INT_Conv2D (weight scale from the weights: init_step_size; two scales), fc_hid / fc_out -> INT_Linear;
    # the functional tanh calls stay float
    ("quant_rvtdcnn_h12_w8a8", "rvtdcnn", 12, 8, 0, 0, False),
    ("quant_rvtdcnn_h6_w16a16", "rvtdcnn", 6, 16, 0, 0, False),
    ("quant_rvtdcnn_h32_w8a8", "rvtdcnn", 32, 8, 0, 0, False),
    # pgjanet: its six nn.Linear (the gates of the cell and the read-out) -> INT_Linear; functional tanh / sigmoid stay float
    ("quant_pgjanet_h11_w8a8", "pgjanet", 11, 8, 0, 0, False),
    ("quant_pgjanet_h9_w16a16", "pgjanet", 9, 16, 0, 0, False),
    ("quant_pgjanet_h24_w8a8", "pgjanet", 24, 8, 0, 0, False),
    # the four backbones whose gates / FIR banks / read-outs are nn.Linear (mcldnn: + two nn.Conv2d) INSIDE a recurrent cell: every one of them is
    # swapped (r05: served by the ATen restatement of the quantised model, opendpd_amd/quant.py::_quantise_aten)
    ("quant_bojanet_h12_w8a8", "bojanet", 12, 8, 0, 0, False),
    ("quant_bojanet_h16_w16a16", "bojanet", 16, 16, 0, 0, False),
    ("quant_apnrru_h8_w8a8", "apnrru", 8, 8, 0, 0, False),
    ("quant_apnrru_h12_w16a16", "apnrru", 12, 16, 0, 0, False),
    ("quant_dvrjanet_h12_w8a8", "dvrjanet", 12, 8, 0, 0, False),
    ("quant_dvrjanet_h10_w16a16", "dvrjanet", 10, 16, 0, 0, False),
    ("quant_mcldnn_h8_w8a8", "mcldnn", 8, 8, 0, 0, False),
    ("quant_mcldnn_h6_w16a16", "mcldnn", 6, 16, 0, 0, False),
]


class P:  # the attributes get_quant_model reads from the Project object (quant/__init__.py:12-18)
    quant = True
    pretrained_model = ""
    quant_dir_label = ""


def main(only):
    x, tgt = gg.real_frames("DPA_200MHz", 5, 37, seed=4)
    xa, ta = gg.real_frames("APA_200MHz", 4, 200, seed=2)
    for name, bb, H, bits, thx, thh, pre in CASES:
        if only and name not in only:
            continue
        fnet = gg.build(bb, H, seed=0, thx=thx, thh=thh, num_dvr_units=3 if bb == "dvrjanet" else None)
        P.n_bits_w = P.n_bits_a = bits
        P.pretrained_model = ""
        extra = {}
        if pre:
            # a float model "trained" elsewhere: another seed's initial weights moved by a few optimiser steps
            src = gg.build(bb, H, seed=11, thx=thx, thh=thh, num_dvr_units=3 if bb == "dvrjanet" else None)
            gg.step_case(src, x, tgt, n_steps=3)
            tmp = tempfile.NamedTemporaryFile(suffix=".pt", delete=False)
            torch.save(src.state_dict(), tmp.name)
            P.pretrained_model = tmp.name
            extra.update(gg.sd_np(src, "pre"))
        torch.manual_seed(123)  # the surgery consumes RNG (INT_Linear draws a fresh bias, _reset_pygru re-initialises)
        qnet = quant.get_quant_model(P, fnet)
        assert qnet is not fnet, "quantisation fell back to the float model"
        extra["rng_after"] = torch.rand(4).numpy()           # the generator state the following train_dpd would see
        if pre:
            os.unlink(P.pretrained_model)
        d = {"x": x, "tgt": tgt, "meta": np.array(json.dumps(
            {"backbone": bb, "hidden": H, "bits": bits, "thx": thx, "thh": thh, "lr": gg.LR, "clip": gg.CLIP, "pretrained": pre,
             "n_param": int(sum(p.numel() for p in qnet.parameters())), "num_dvr_units": 3 if bb == "dvrjanet" else None}))}
        d.update(extra)
        d.update(gg.sd_np(fnet, "fsd"))                      # the float model the surgery started from
        d.update(gg.sd_np(qnet, "sd"))
        qnet.eval()
        with torch.no_grad():
            gg.reset_stats(qnet)
            d["y_eval"] = qnet(torch.from_numpy(x)).numpy().copy()
            gg.reset_stats(qnet)
            d["xa"], d["ta"] = xa, ta
            d["ya_eval"] = qnet(torch.from_numpy(xa)).numpy().copy()
            sa = gg.read_stats(qnet)
            if sa:
                d["stats_a"] = sa["stats"]
        d.update(gg.step_case(qnet, x, tgt))
        qnet.train()
        with torch.no_grad():
            d["y_p3_train"] = qnet(torch.from_numpy(x)).numpy().copy()
        qnet.eval()
        with torch.no_grad():
            d["y_p3_eval"] = qnet(torch.from_numpy(x)).numpy().copy()
        d.update(gg.sd_np(qnet, "sd3"))
        gg.save(name, d)


if __name__ == "__main__":
    main(sys.argv[1:])
