"""BASELINE-size checks (65 536 frames x 200, the bench workload) through size-independent properties — the oracle finishes such a batch
in minutes, not seconds, so at this size the HIP path is checked by: additivity of loss and gradient over a split of the batch,
invariance under a permutation of the frames, agreement of the two kernel mappings, bit-repeatability, and an oracle spot check of
randomly drawn frames of the full-size forward pass."""
import numpy as np
import pytest
import torch

from tests.golden_util import rel_err

pytestmark = pytest.mark.gpu
B, T, H = 65536, 200, 13


@pytest.fixture(scope="module")
def workload():
    import bench
    from opendpd_amd import CoreModel
    dev = torch.device("cuda:0")
    xs, ys = bench.synth_frames(B, T, 0, dev, materialize=False)      # the bench's synthetic APA_200MHz-shaped streams
    torch.manual_seed(0)
    net = CoreModel(2, H, 1, "dgru").to(dev)
    return net, xs, ys


def _grad(net, xs, ys, order, count):
    """loss sum and gradient of the fused train step on the frames `order` of the streams, normalised by `count` elements"""
    from opendpd_amd.train_funcs import FrameBatch, FusedAdamW, fused_train_step
    p0 = net.backbone.flat_params().clone()
    opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
    loss = fused_train_step(opt, FrameBatch(xs, ys, order, T, 1), None, "l2", 0.0, global_count=count)
    assert torch.equal(net.backbone.flat_params(), p0)               # lr 0: the parameters did not move
    return float(loss), opt.grad[:-4].clone(), opt


def test_gradient_and_loss_are_additive_over_the_batch_and_permutation_invariant(workload):
    net, xs, ys = workload
    dev = xs.device
    count = B * T * 2
    full = torch.arange(B, device=dev)
    l, g, opt = _grad(net, xs, ys, full, count)
    assert opt.train_workspace(B, T, dev) is not None                 # the 16-sequences-per-wave kernel serves this size
    cut = 40001                                                       # ragged split: neither part a multiple of 16
    la, ga, _ = _grad(net, xs, ys, full[:cut].contiguous(), count)
    lb, gb, _ = _grad(net, xs, ys, full[cut:].contiguous(), count)
    assert abs((la + lb) - l) < 2e-6 * abs(l)
    assert rel_err((ga + gb).cpu().numpy(), g.cpu().numpy()) < 2e-5
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(dev)
    lp, gp, _ = _grad(net, xs, ys, perm, count)
    assert abs(lp - l) < 2e-6 * abs(l)
    assert rel_err(gp.cpu().numpy(), g.cpu().numpy()) < 2e-5
    l2, g2, _ = _grad(net, xs, ys, full, count)                       # same launch twice: bit-identical
    assert l2 == l and torch.equal(g2, g)


def test_kernel_mappings_agree_at_full_size(workload):
    """the S16 (MFMA) kernel against the row-rotated (DPP) kernel on the same 65 536 frames: different summation orders, and a relu
    pre-activation within rounding of 0 may take a different mask — gradients agree to 2e-3 of their maximum"""
    from opendpd_amd import _lib
    net, xs, ys = workload
    lib = _lib.load()
    order = torch.arange(B, device=xs.device)
    try:
        lib.odpd_set_tuning(b"s16_min_batch", -1)
        l16, g16, _ = _grad(net, xs, ys, order, B * T * 2)
        lib.odpd_set_tuning(b"s16_min_batch", 1 << 40)
        lrr, grr, opt = _grad(net, xs, ys, order, B * T * 2)
        assert opt.train_workspace(B, T, xs.device) is None            # row-rotated: BPTT state in LDS, no workspace
    finally:
        lib.odpd_set_tuning(b"s16_min_batch", -1)
    assert abs(l16 - lrr) < 2e-6 * abs(lrr)
    assert rel_err(g16.cpu().numpy(), grr.cpu().numpy()) < 2e-3


def test_full_size_forward_against_the_oracle_on_drawn_frames(workload):
    from oracle.oracle import Oracle, make_model
    net, xs, ys = workload
    x = xs.unfold(0, T, 1)[:B].permute(0, 2, 1).contiguous()         # the (B,T,2) frames IQFrameDataset would materialise
    with torch.no_grad():
        y = net(x)
    assert y.shape == (B, T, 2) and bool(torch.isfinite(y).all())
    idx = np.sort(np.random.RandomState(0).choice(B, 48, replace=False))
    p = net.backbone.flat_params().detach().cpu().numpy()
    yo, _ = Oracle("f32").forward(make_model("dgru", H), p, x[idx].cpu().numpy())
    assert rel_err(y[idx].cpu().numpy(), yo) < 2e-5
    # checksum of checksums: the mean over the batch equals the size-weighted mean over a ragged split of it
    m = float(y.double().mean())
    cut = 12345
    assert abs((float(y[:cut].double().sum()) + float(y[cut:].double().sum())) / y.numel() - m) < 1e-9


def test_cascade_gradient_is_additive_at_config_3_size():
    """BASELINE config 3 shape at a saturating batch (TRes-DeltaGRU H15 with its thresholds -> frozen DGRU H23, 16 384 x 200): the DPD
    gradient of the train_dpd step is the sum of the gradients of a ragged split (thresholded deltas act per sequence)"""
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    Bc = 16384
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.rand(Bc, T, 2, device="cuda", generator=g) - 0.5) * 1.2
    x = x + 0.05 * torch.sign(x)
    torch.manual_seed(0)
    net = CascadedModel(dpd_model=CoreModel(2, 15, 1, "deltagru_tcnskip", thx=0.01, thh=0.05), pa_model=CoreModel(2, 23, 1, "dgru"))
    net.freeze_pa_model()
    net = net.cuda()
    count = Bc * T * 2

    def grad(sl):
        opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
        xb = x[sl].contiguous()
        loss = fused_train_step(opt, xb, xb.clone(), "l2", 0.0, global_count=count)
        return float(loss), opt.grad[:-4].clone()
    l, gfull = grad(slice(0, Bc))
    la, ga = grad(slice(0, 9001))
    lb, gb = grad(slice(9001, Bc))
    assert abs((la + lb) - l) < 5e-6 * abs(l)
    assert rel_err((ga + gb).cpu().numpy(), gfull.cpu().numpy()) < 5e-5
