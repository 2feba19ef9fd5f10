"""bf16 storage of the resident I/Q streams (BASELINE configs[1] "bf16"; odpd_frames_t.sample_format = ODPD_SAMPLES_BF16, opt-in through
`frame_storage="bf16"`).  Declared behaviour, checked here:
  * the kernels widen every stored value exactly and compute in fp32, so a step on bf16 streams is BIT-IDENTICAL to the same step on fp32
    streams holding the bf16-rounded values — in every kernel regime of the GRU family (one frame per wave, four per wave, sixteen per
    wave, hidden 17..32);
  * against the fp32 oracle run on those rounded values the usual fp32 tolerance holds (2e-5 on the loss, 1e-3 relative on gradients);
  * against the UNROUNDED data the inputs carry a relative rounding of 2^-9: the first-epoch TRAIN_LOSS of an APA-like run moves by less
    than 2 %, the validation NMSE by less than 0.3 dB (the tolerance declared in include/opendpd_hip.h / DESIGN);
  * backbones whose kernels do not read the format refuse it (ODPD_EUNSUPPORTED) instead of misreading the stream."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _streams(n, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(n, 2, generator=g) - 0.5) * 1.4
    x = x + 0.05 * torch.sign(x)
    y = x * (1.0 - 0.2 * (x * x).sum(-1, keepdim=True)) + 0.05 * torch.roll(x, 1, 0)
    return x.cuda().contiguous(), y.cuda().contiguous()


@pytest.fixture
def s16_everywhere():
    from opendpd_amd import _lib
    lib = _lib.load()
    yield lambda on: lib.odpd_set_tuning(b"s16_min_batch", 0 if on else -1)
    lib.odpd_set_tuning(b"s16_min_batch", -1)


@pytest.mark.parametrize("bb,H,B,T,s16", [("dgru", 13, 64, 200, False),      # one frame per wave (gate-parallel)
                                          ("dgru", 13, 2000, 50, False),     # four frames per wave (row-rotated)
                                          ("gru", 11, 100, 37, True),        # sixteen per wave (S16), ragged
                                          ("dgru", 13, 333, 50, True),
                                          ("qgru", 10, 64, 50, True),
                                          ("dgru", 23, 100, 50, True),       # hidden 17..32 (S16N)
                                          ("gru", 30, 64, 40, True)])
def test_bf16_streams_equal_fp32_streams_of_the_rounded_values(bb, H, B, T, s16, s16_everywhere):
    from opendpd_amd import CoreModel
    from opendpd_amd.train_funcs import FrameBatch, FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    s16_everywhere(s16)
    x, y = _streams(B + T + 40, 4)
    xb, yb = x.to(torch.bfloat16), y.to(torch.bfloat16)
    order = torch.randperm(B + 40, generator=torch.Generator().manual_seed(1))[:B].cuda()
    out = []
    for xs, ys in ((xb, yb), (xb.float(), yb.float())):
        torch.manual_seed(0)
        net = CoreModel(2, H, 1, bb).cuda()
        opt = FusedAdamW(net, lr=1e-3)
        loss = fused_train_step(opt, FrameBatch(xs, ys, order, T, 1), None, "l2", 200.0)
        torch.cuda.synchronize()
        out.append((float(loss.item()), opt.grad.cpu().numpy().copy(), net.backbone.flat_params().cpu().numpy().copy()))
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    if H <= 16 and B <= 400:      # and the oracle on the rounded values (small cases: it is a scalar CPU port)
        torch.manual_seed(0)
        net = CoreModel(2, H, 1, bb)
        p = np.concatenate([q.detach().numpy().reshape(-1) for q in net.parameters()])
        idx = order.cpu().numpy()
        fx = np.stack([xb.float().cpu().numpy()[i:i + T] for i in idx])
        fy = np.stack([yb.float().cpu().numpy()[i:i + T] for i in idx])
        o, m = Oracle("f32"), make_model(bb, H)
        yo, _ = o.forward(m, p, fx)
        lo, dy = o.loss("l2", yo, fy)
        go, _ = o.backward(m, p, fx, dy, need_dx=False)
        assert abs(out[0][0] - lo) <= 2e-5 * max(1.0, abs(lo))
        assert np.abs(out[0][1][:len(go)] - go).max() <= 1e-3 * np.abs(go).max()


def test_backbones_that_do_not_read_the_format_refuse_it():
    from opendpd_amd import CoreModel
    from opendpd_amd.train_funcs import FrameBatch, FusedAdamW, fused_train_step
    x, y = _streams(400, 2)
    fb = FrameBatch(x.to(torch.bfloat16), y.to(torch.bfloat16), torch.arange(64).cuda(), 50, 1)
    net = CoreModel(2, 14, 1, "lstm").cuda()
    with pytest.raises(RuntimeError, match="unsupported"):
        fused_train_step(FusedAdamW(net, lr=1e-3), fb, None, "l2", 200.0)
    with pytest.raises(TypeError):
        FrameBatch(x.to(torch.bfloat16), y, torch.arange(64).cuda(), 50, 1)


@pytest.fixture
def workdir(tmp_path_factory):
    wd = tmp_path_factory.mktemp("odpd_bf16")
    d = dict(np.load(os.path.join(GOLDEN, "dpa200_dataset.npz")))
    ds = wd / "datasets" / "DPA_200MHz"
    ds.mkdir(parents=True)
    (ds / "spec.json").write_text(str(d.pop("spec")))
    for k, v in d.items():
        pd.DataFrame(v, columns=["I", "Q"]).to_csv(ds / f"{k}.csv", index=False)
    old = os.getcwd()
    os.chdir(wd)
    os.environ["OPENDPD_DATASETS"] = str(wd / "datasets")
    yield wd
    os.chdir(old)


@pytest.mark.parametrize("bb,H", [("dgru", 13), ("lstm", 14)])
def test_train_pa_with_bf16_frame_storage_stays_within_the_declared_tolerance(workdir, bb, H):
    """dgru: the native epoch loop reads the bf16 streams in place; lstm: its kernels take fp32 batches, the loader widens the gathered
    frames (same rounded data, generic path).  Both against the fp32-storage run of the same seed."""
    import opendpd_amd as od
    hist = {}
    for storage in ("fp32", "bf16"):
        res = od.train_pa(dataset_name="DPA_200MHz", PA_backbone=bb, PA_hidden_size=H, frame_length=50, batch_size=64, lr=1e-3, n_epochs=1,
                          seed=0, accelerator="cuda", frame_storage=storage)
        assert res["status"] == "completed"
        hist[storage] = pd.read_csv(os.path.join("log", "DPA_200MHz", "train_pa", "history", os.path.basename(res["log_path"])))
    a, b = hist["fp32"], hist["bf16"]
    assert abs(b["TRAIN_LOSS"][0] - a["TRAIN_LOSS"][0]) <= 0.02 * a["TRAIN_LOSS"][0]
    assert abs(b["VAL_NMSE"][0] - a["VAL_NMSE"][0]) <= 0.3 and abs(b["TEST_NMSE"][0] - a["TEST_NMSE"][0]) <= 0.3
    assert b["TRAIN_LOSS"][0] != a["TRAIN_LOSS"][0]          # (the storage really was different)
