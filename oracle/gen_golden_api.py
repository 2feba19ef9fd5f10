#!/usr/bin/env python3
"""Wire-format vectors of opendpd.api.create_dataset (TEST INFRASTRUCTURE — build container only): runs the reference's function
(opendpd/api.py:316-431, loaded from /root/reference) on a small seeded CSV for both layouts and stores every file it wrote as
text, plus what its load_dataset returns (tests/golden/create_dataset_ref.json); also the argparse defaults of arguments.py
(tests/golden/argument_defaults.json) and the call signatures of the public API (tests/golden/api_signatures.json).  Usage: python oracle/gen_golden_api.py"""
import importlib.util
import json
import os
import sys
import tempfile

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
KW = dict(train_ratio=0.55, val_ratio=0.25, test_ratio=0.2, input_signal_fs=800e6, bw_main_ch=200e6, n_sub_ch=10)


def main():
    spec = importlib.util.spec_from_file_location("ref_api", os.path.join(REF, "opendpd", "api.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.RandomState(0)
    df = pd.DataFrame(rng.randn(23, 4), columns=["I_in", "Q_in", "I_out", "Q_out"])
    out = {"csv": df.to_csv(index=False), "kwargs": KW, "cases": {}}
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "in.csv"), "w").write(out["csv"])
        for name, extra in (("single_csv", {}), ("split_csv", {}), ("single_named", {"dataset_format": "SINGLE_CSV", "csv_filename": "pa.csv"}),
                            ("defaults", None)):
            kw = {**dict(dataset_format=name), **KW, **extra} if extra is not None else {}
            d = ref.create_dataset(os.path.join(tmp, "in.csv"), os.path.join(tmp, name), "MyPA", **kw)
            loaded = ref.load_dataset(d)
            out["cases"][name] = {"kwargs": kw, "files": {f: open(os.path.join(d, f)).read() for f in sorted(os.listdir(d))},
                                  "loaded": {k: np.asarray(v).tolist() for k, v in loaded.items()}}
    json.dump(out, open(os.path.join(OUT, "create_dataset_ref.json"), "w"), indent=1)
    # utils/metrics.py on seeded random segments (inputs are re-drawn from the seed by the test): exact values as float.hex()
    from utils import metrics as rm
    cases = []
    for trial in range(24):
        rng = np.random.RandomState(1000 + trial)
        nseg, n = int(rng.randint(1, 4)), int(rng.choice([256, 512, 1000, 2560]))
        nperseg = int(rng.choice([64, 128, 256, n])) if trial % 2 else n
        dt = "float32" if trial % 3 else "float64"
        pred = (rng.randn(nseg, n, 2) * 0.3).astype(dt)
        truth = (pred + 0.05 * rng.randn(nseg, n, 2)).astype(dt)
        fs, bw, nsub = float(rng.choice([800e6, 983.04e6])), float(rng.choice([100e6, 200e6])), int(rng.choice([1, 2, 5, 10]))
        if n != nperseg:
            vals = (rm.NMSE(pred, truth), *rm.ACLR(pred, fs=fs, nperseg=nperseg, bw_main_ch=bw, n_sub_ch=nsub))
        else:
            vals = (rm.NMSE(pred, truth), *rm.ACLR(pred, fs=fs, nperseg=nperseg, bw_main_ch=bw, n_sub_ch=nsub),
                    rm.EVM(pred, truth, bw_main_ch=bw, n_sub_ch=nsub, nperseg=nperseg))
        cases.append({"seed": 1000 + trial, "trial": trial, "fs": fs, "bw": bw, "nsub": nsub, "nperseg": nperseg,
                      "values": [float(v).hex() for v in vals], "types": [type(v).__name__ for v in vals]})
    json.dump(cases, open(os.path.join(OUT, "metrics_exact.json"), "w"), indent=1)
    # call signatures of the public API (opendpd/api.py): parameter names, order, defaults, kinds
    import inspect
    sig = lambda f: [[n, None if q.default is inspect._empty else q.default, str(q.kind)] for n, q in inspect.signature(f).parameters.items()]
    sigs = {n: sig(getattr(ref, n)) for n in ("train_pa", "train_dpd", "run_dpd", "load_dataset", "create_dataset")}
    sigs.update({"OpenDPDTrainer." + n: sig(getattr(ref.OpenDPDTrainer, n)) for n in ("__init__", "train_pa", "train_dpd", "run")})
    sigs["OpenDPDTrainer.public"] = [n for n in dir(ref.OpenDPDTrainer) if not n.startswith("_")]
    json.dump(sigs, open(os.path.join(OUT, "api_signatures.json"), "w"), indent=1)
    # argparse defaults of the reference's CLI (arguments.py:8-89), what `Project` falls back to
    sys.argv = ["main.py"]
    import arguments
    json.dump(vars(arguments.get_arguments()), open(os.path.join(OUT, "argument_defaults.json"), "w"), indent=1, sort_keys=True)
    print({k: sorted(v["files"]) for k, v in out["cases"].items()})


if __name__ == "__main__":
    main()
