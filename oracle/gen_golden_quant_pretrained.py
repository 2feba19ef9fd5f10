#!/usr/bin/env python3
"""Anchor for `--pretrained_model` in the quantised flow (TEST INFRASTRUCTURE — build container only).

Runs the REFERENCE's `quant.get_quant_model` (quant/__init__.py:20-37 -> Base_GRUQuantEnv.load_model,
quant_envs.py:173-182) with the three kinds of checkpoint a user can hand it and records what comes out:

  pygru  a float checkpoint with the PyGRU key names (`backbone.rnn.rnn_cell_list.0.{x2h,h2h}.{weight,bias}`, `backbone.fc_out.*`):
         strict-loaded into the float holder BEFORE quantisation; INT_Linear then keeps only the weights and draws fresh
         biases (quant_layers.py:48-56) -> the quantised state dict is stored;
  nngru  the checkpoint a float `train_dpd --DPD_backbone qgru` writes (`backbone.rnn.weight_ih_l0` ...): the strict load raises,
         get_quant_model prints "[WARN] Quantization setup failed" and hands back the FLOAT model object it was given;
  quant  a quantised checkpoint (scales and side-effect buffers among the keys): same fallback.

Writes tests/golden/quant_pretrained_qgru_h10.npz.   Usage: python oracle/gen_golden_quant_pretrained.py"""
import json
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import torch  # noqa: E402

torch.set_num_threads(1)
import quant  # noqa: E402
from quant.modules.ops import Sqrt, Pow  # noqa: E402

quant.Sqrt, quant.Pow = Sqrt, Pow      # harness-side bridge for the reference's import defect
import models as ref_models  # noqa: E402

H, BITS = 10, 8


class P:
    quant = True
    n_bits_w = n_bits_a = BITS
    quant_dir_label = ""
    pretrained_model = ""


def float_model(seed):
    torch.manual_seed(seed)
    return ref_models.CoreModel(2, H, 1, "qgru")


def main():
    d = {}
    with tempfile.TemporaryDirectory() as tmp:
        # key names of the float holder: taken from the reference's own environment object
        P.pretrained_model = ""
        torch.manual_seed(123)
        q0 = quant.get_quant_model(P, float_model(0))
        holder_sd = P.quant_env.pygru_model.state_dict()
        g = torch.Generator().manual_seed(7)
        pre = {k: (torch.rand(v.shape, generator=g) - 0.5) * 0.8 for k, v in holder_sd.items()}
        path = os.path.join(tmp, "pygru.pt")
        torch.save(pre, path)
        P.pretrained_model = path
        fnet = float_model(0)
        torch.manual_seed(123)
        q = quant.get_quant_model(P, fnet)
        assert q is not fnet
        for k, v in pre.items():
            d["pre/" + k] = v.numpy()
        for k, v in q.state_dict().items():
            d["sd/" + k] = v.numpy()
        outcomes = {"pygru": "quantised"}
        # the two kinds the strict load refuses
        nn_path = os.path.join(tmp, "nngru.pt")
        torch.save(float_model(1).state_dict(), nn_path)
        q_path = os.path.join(tmp, "quant.pt")
        torch.save(q0.state_dict(), q_path)
        for kind, pth in (("nngru", nn_path), ("quant", q_path)):
            P.pretrained_model = pth
            fnet = float_model(0)
            before = {k: v.clone() for k, v in fnet.state_dict().items()}
            torch.manual_seed(321)
            out = quant.get_quant_model(P, fnet)
            d[f"rng_after_refused_{kind}"] = torch.rand(4).numpy()      # the surgery's RNG draws up to the failing load have happened
            same = out is fnet and all(torch.equal(before[k], v) for k, v in out.state_dict().items())
            outcomes[kind] = "float model returned unchanged" if same else "other"
        # an unreadable checkpoint: torch.load raises inside the same try block
        P.pretrained_model = os.path.join(tmp, "missing.pt")
        fnet = float_model(0)
        torch.manual_seed(321)
        out = quant.get_quant_model(P, fnet)
        outcomes["missing"] = "float model returned unchanged" if out is fnet else "other"
        d["rng_after_refused_missing"] = torch.rand(4).numpy()
        d["nngru_keys"] = np.array(json.dumps(list(torch.load(nn_path).keys())))
    d["meta"] = np.array(json.dumps({"backbone": "qgru", "hidden": H, "bits": BITS, "outcomes": outcomes,
                                     "n_param": int(sum(p.numel() for p in q.parameters()))}))
    np.savez_compressed(os.path.join(OUT, "quant_pretrained_qgru_h10.npz"), **d)
    print(outcomes)


if __name__ == "__main__":
    main()
