#!/usr/bin/env python3
"""Golden vectors for the registry backbones without HIP kernels — TEST INFRASTRUCTURE, runs only in the
build container.  Imports the reference's CoreModel from /root/reference (read-only), seeds it, and stores the initial
state dict, one forward output, the MSE loss against a stored target and the gradients of every parameter and of x
(backbones/{gmp,rvtdcnn,apnrru,bojanet,deltajanet,dvrjanet,neuraltx,mcldnn}.py through models.py:26-148).

Usage:  python oracle/gen_golden_extras.py      (writes tests/golden/extra_*.npz)
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import torch  # noqa: E402

torch.set_num_threads(1)
import models as ref_models  # noqa: E402

CASES = [("gmp", 8), ("rvtdcnn", 6), ("apnrru", 8), ("bojanet", 8), ("bojanet", 15), ("deltajanet", 10), ("dvrjanet", 8),
         ("neuraltx", 12), ("mcldnn", 8)]


def main():
    for bb, H in CASES:
        torch.manual_seed(0)
        net = ref_models.CoreModel(2, H, 1, bb, window_size=4, num_dvr_units=4, thx=0.01, thh=0.05)
        after = float(torch.rand(1))             # position of the global RNG after construction
        sd = {k: v.detach().clone().numpy() for k, v in net.state_dict().items()}
        rng = np.random.RandomState(7)
        amp = 0.05 + 0.85 * rng.rand(3, 40, 1)
        ph = 2 * np.pi * rng.rand(3, 40, 1)
        x = np.concatenate([amp * np.cos(ph), amp * np.sin(ph)], -1).astype(np.float32)
        tgt = (0.5 * rng.randn(3, 40, 2)).astype(np.float32)
        with torch.no_grad():                   # biases are zero after most inits: make every parameter count
            for k, p in net.named_parameters():
                if "bias" in k or k.endswith(".Z"):
                    p.copy_(torch.from_numpy(rng.uniform(-0.3, 0.3, tuple(p.shape)).astype(np.float32)))
        sd_used = {k: v.detach().clone().numpy() for k, v in net.state_dict().items()}
        xt = torch.from_numpy(x).requires_grad_(True)
        y = net(xt)
        loss = torch.nn.functional.mse_loss(y, torch.from_numpy(tgt))
        loss.backward()
        out = {"meta": json.dumps({"backbone": bb, "hidden": H, "n_param": int(sum(p.numel() for p in net.parameters())),
                                   "rng_after_init": after, "loss": float(loss)}),
               "x": x, "tgt": tgt, "y": y.detach().numpy(), "gx": xt.grad.numpy()}
        for k, v in sd.items():
            out["sd/" + k] = v
        for k, v in sd_used.items():
            out["sdu/" + k] = v
        for k, p in net.named_parameters():
            out["g/" + k] = p.grad.numpy()
        np.savez_compressed(os.path.join(OUT, f"extra_{bb}_h{H}.npz"), **out)
        print(bb, H, "loss", float(loss), "params", json.loads(out["meta"])["n_param"])


if __name__ == "__main__":
    main()
