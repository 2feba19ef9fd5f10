#!/usr/bin/env python3
"""20 train steps of the quantised pgjanet (W8A8, hidden $EXP_H, 256 x 200) for a rocprofv3 --kernel-trace --stats run."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step  # noqa: E402
from tests.test_quant_more_gpu import _fresh  # noqa: E402

dev = torch.device("cuda:0")
x, t = bench.synth_frames(256, 200, 0, dev)
net = _fresh("pgjanet", int(os.environ.get("EXP_H", "11")), 8).to(dev)
opt = FusedAdamW(net, lr=5e-4)
for _ in range(20):
    loss = fused_train_step(opt, x, t, "l2", 200.0, 256 * 200 * 2)
torch.cuda.synchronize()
print(float(loss))
