#!/usr/bin/env python3
"""Forward parity on the evaluation shapes of the bundled datasets (net_eval / run_dpd: (3, 2560, 2) and (1, 19662, 2) segments,
train_funcs.py:57-90): every HIP backbone against the float64 oracle.  usage: PYTHONPATH=. python tools/eval_parity.py"""
import warnings

import numpy as np
import torch

from opendpd_amd import CoreModel
from oracle.oracle import Oracle, make_model

o = Oracle("f64")
o.set_threads(o.max_threads())
for bb, H, kw in (("gru", 11, {}), ("dgru", 13, {}), ("dgru", 23, {}), ("qgru", 10, {}), ("qgru_amp1", 16, {}), ("lstm", 14, {}), ("vdlstm", 13, {}),
                  ("deltagru", 15, dict(thx=0.0, thh=0.0)), ("deltagru_tcnskip", 15, dict(thx=0.0, thh=0.0)), ("deltagru_tcnskip", 24, dict(thx=0.0, thh=0.0)),
                  ("pgjanet", 11, {}), ("tcnn", 35, {}), ("gmp", 11, {}), ("rvtdcnn", 25, {}), ("neuraltx", 36, {}), ("deltajanet", 15, {}),
                  ("dvrjanet", 12, dict(num_dvr_units=3)), ("bojanet", 12, {}), ("apnrru", 8, {}), ("mcldnn", 8, {})):
    for B, T in ((3, 2560), (1, 19662)):
        torch.manual_seed(0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net = CoreModel(2, H, 1, bb, **kw).cuda().eval()
        if bb == "apnrru":          # Z = 0 at construction switches the deep cell off
            with torch.no_grad():
                net.backbone.rru.Z.uniform_(-0.5, 0.5)
        g = torch.Generator().manual_seed(B + T)
        amp, ph = 0.05 + 0.85 * torch.rand(B, T, 1, generator=g), 2 * np.pi * torch.rand(B, T, 1, generator=g)
        x = torch.cat((amp * torch.cos(ph), amp * torch.sin(ph)), -1)
        p = np.concatenate([q.detach().cpu().numpy().reshape(-1) for q in net.parameters()])
        yo, _ = o.forward(make_model(bb, H, kw.get("thx", 0), kw.get("thh", 0), bits_w=kw.get("num_dvr_units", 0)), p.astype(np.float64),
                          x.numpy().astype(np.float64))
        with torch.no_grad():
            y = net(x.cuda()).cpu().numpy()
        e = np.abs(y - yo) / np.abs(yo).max()
        print(f"{bb:18s} H{H:<3d} ({B}, {T}, 2): max rel err {e.max():.2e}  (last 100 steps {e[:, -100:].max():.2e})", flush=True)
