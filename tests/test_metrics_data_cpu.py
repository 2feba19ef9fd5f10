"""Host metrics and framing against known answers produced by the reference (tests/golden/metrics_framing.npz and,
when the reference checkout with its bundled datasets is present, the dataset-level numbers of BASELINE.md §2)."""
import json
import os

import numpy as np
import pytest
import torch

from opendpd_amd import data, metrics
from tests.golden_util import GOLDEN


def _g():
    return dict(np.load(os.path.join(GOLDEN, "metrics_framing.npz")))


def test_metrics_known_answers():
    g = _g()
    assert abs(metrics.NMSE(g["pred"], g["truth"]) - float(g["NMSE"])) < 1e-6
    assert abs(metrics.EVM(g["pred"], g["truth"], bw_main_ch=200e6, n_sub_ch=2, nperseg=512) - float(g["EVM"])) < 1e-6
    al, ar = metrics.ACLR(g["pred"], fs=800e6, nperseg=512, bw_main_ch=200e6, n_sub_ch=2)
    assert np.allclose([al, ar], g["ACLR"], atol=1e-6)


def test_framing_and_segments():
    g = _g()
    for s in (1, 7):
        f = data.frames(g["stream"], 50, s)
        assert list(f.shape) == list(g[f"frames_s{s}_shape"])
        assert np.array_equal(f[0].numpy(), g[f"frames_s{s}_first"]) and np.array_equal(f[-1].numpy(), g[f"frames_s{s}_last"])
    assert np.array_equal(data.segments(g["stream"], 128).numpy(), g["segs"])
    torch.manual_seed(0)
    assert np.array_equal(torch.randperm(22991)[:16].numpy(), g["perm_head_22991"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/datasets/DPA_200MHz"), reason="reference datasets not present")
def test_dataset_level_known_answers():
    ans = json.load(open(os.path.join(GOLDEN, "dataset_known_answers.json")))
    os.environ["OPENDPD_DATASETS"] = "/root/reference/datasets"
    for ds, ref in ans.items():
        spec = data.load_spec(dataset_name=ds)
        Xtr, ytr, _, _, Xte, yte = data.load_dataset(dataset_name=ds)
        g = data.set_target_gain(Xtr, ytr)
        assert abs(g - ref["target_gain"]) < 1e-12
        pred = data.segments(yte, spec["nperseg"]).numpy()
        truth = data.segments(g * Xte, spec["nperseg"]).numpy()
        assert list(pred.shape) == ref["seg_shape"]
        assert abs(metrics.NMSE(pred, truth) - ref["NMSE"]) < 1e-5
        assert abs(metrics.EVM(pred, truth, bw_main_ch=spec["bw_main_ch"], n_sub_ch=spec["n_sub_ch"], nperseg=spec["nperseg"]) - ref["EVM"]) < 1e-5
        al, ar = metrics.ACLR(pred, fs=spec["input_signal_fs"], nperseg=spec["nperseg"], bw_main_ch=spec["bw_main_ch"],
                              n_sub_ch=spec["n_sub_ch"])
        assert abs(al - ref["ACLR_L"]) < 1e-5 and abs(ar - ref["ACLR_R"]) < 1e-5   # float32 spectra


def test_metrics_are_bit_identical_with_the_reference():
    """utils/metrics.py:42-187 on seeded random segments (float32 and float64 inputs, 1..10 sub-channels, Welch segments shorter
    than the record): the same doubles and the same numpy scalar types as the REFERENCE's functions returned
    (tests/golden/metrics_exact.json, oracle/gen_golden_api.py) — the history CSV prints them with 8 decimals."""
    import json
    import os
    from opendpd_amd import metrics as M
    from tests.golden_util import GOLDEN
    for c in json.load(open(os.path.join(GOLDEN, "metrics_exact.json"))):
        trial = c["trial"]
        rng = np.random.RandomState(c["seed"])
        nseg, n = int(rng.randint(1, 4)), int(rng.choice([256, 512, 1000, 2560]))
        nperseg = int(rng.choice([64, 128, 256, n])) if trial % 2 else n
        dt = "float32" if trial % 3 else "float64"
        pred = (rng.randn(nseg, n, 2) * 0.3).astype(dt)
        truth = (pred + 0.05 * rng.randn(nseg, n, 2)).astype(dt)
        assert nperseg == c["nperseg"]
        vals = [M.NMSE(pred, truth), *M.ACLR(pred, fs=c["fs"], nperseg=nperseg, bw_main_ch=c["bw"], n_sub_ch=c["nsub"])]
        if n == nperseg:
            vals.append(M.EVM(pred, truth, bw_main_ch=c["bw"], n_sub_ch=c["nsub"], nperseg=nperseg))
        assert [float(v).hex() for v in vals] == c["values"], c
        assert [type(v).__name__ for v in vals] == c["types"], c


def test_calculate_metrics_many_equals_per_run_calls_bit_for_bit():
    """sweep-level metrics (opendpd_amd/sweep.py): one FFT / Welch call over the K runs' stacked predictions gives every run the numbers — and
    the number TYPES (they reach the CSV formatter) — of its own calculate_metrics call (utils/metrics.py:42-187 of the reference)"""
    from types import SimpleNamespace
    from opendpd_amd.metrics import calculate_metrics, calculate_metrics_many
    args = SimpleNamespace(nperseg=2560, n_sub_ch=10, bw_main_ch=200e6, input_signal_fs=800e6)
    rng = np.random.RandomState(3)
    K, S, N = 5, 4, 2560
    truth = (0.3 * rng.randn(S, N, 2)).astype(np.float32)
    preds = [(truth + 10.0 ** -(1 + k) * rng.randn(S, N, 2)).astype(np.float32) for k in range(K)]
    for truths in ([truth] * K, [(truth * (1 + 0.01 * k)).astype(np.float32) for k in range(K)]):
        one = [calculate_metrics(args, {"loss": 0.0}, p, g) for p, g in zip(preds, truths)]
        many = calculate_metrics_many(args, [{"loss": 0.0} for _ in range(K)], preds, truths)
        for a, b in zip(one, many):
            assert list(a.keys()) == list(b.keys())
            for k in a:
                assert a[k] == b[k] and type(a[k]) is type(b[k]), (k, a[k], b[k])
    # ragged shapes fall back to the per-run function
    many = calculate_metrics_many(args, [{}, {}], [preds[0], preds[1][:2]], [truth, truth[:2]])
    assert many[1]["NMSE"] == calculate_metrics(args, {}, preds[1][:2], truth[:2])["NMSE"]
