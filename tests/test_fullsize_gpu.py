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


def test_full_size_gradient_against_the_oracle_on_drawn_frames(workload):
    """The GRADIENT of the full-size launch against the oracle (only additivity and a forward spot check reached this size).
    256 drawn frames, each placed 256 times at shuffled positions of a 65 536-frame batch: the launch is the bench's (same grid, same
    16-sequences-per-wave kernel, same reduction over 256 partial rows), its loss and gradient are the mean over identical copies — i.e.
    the oracle's loss and gradient of the 256 drawn frames."""
    from oracle.oracle import Oracle, make_model
    net, xs, ys = workload
    dev = xs.device
    rng = np.random.RandomState(3)
    idx = np.sort(rng.choice(B, 256, replace=False))
    order = torch.from_numpy(rng.permutation(np.repeat(idx, B // 256))).to(dev)
    l, g, opt = _grad(net, xs, ys, order, B * T * 2)
    assert opt.train_workspace(B, T, dev) is not None                 # the 16-sequences-per-wave kernel served it
    x = xs.unfold(0, T, 1).permute(0, 2, 1)[torch.from_numpy(idx).to(dev)].contiguous().cpu().numpy()
    t = ys.unfold(0, T, 1).permute(0, 2, 1)[torch.from_numpy(idx).to(dev)].contiguous().cpu().numpy()
    o = Oracle("f32")
    m = make_model("dgru", H)
    p = net.backbone.flat_params().detach().cpu().numpy()
    y, _ = o.forward(m, p, x)
    lo, dy = o.loss("l2", y, t)
    go, _ = o.backward(m, p, x, dy, need_dx=False)
    assert abs(l - lo) < 2e-6 * abs(lo)
    assert rel_err(g.cpu().numpy(), go) < 3e-4                        # the small-shape gradient tolerance of tests/test_gru_family_gpu.py


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


_EVERY = [("gru", 11, {}), ("dgru", 13, {}), ("dgru", 23, {}), ("qgru", 10, {}), ("qgru_amp1", 10, {}), ("lstm", 14, {}), ("vdlstm", 13, {}),
          ("deltagru", 15, dict(thx=0.01, thh=0.05)), ("deltagru_tcnskip", 15, dict(thx=0.01, thh=0.05)), ("pgjanet", 11, {}), ("tcnn", 35, {}),
          ("gmp", 11, {}), ("rvtdcnn", 25, {}), ("neuraltx", 36, {}), ("deltajanet", 15, {}), ("dvrjanet", 12, dict(num_dvr_units=3)),
          ("bojanet", 12, {}), ("apnrru", 8, {}), ("mcldnn", 8, {}), ("qgru W8A8", 10, {}), ("qgru_amp1 W8A8", 10, {})]


@pytest.mark.parametrize("bb,hidden,kw", _EVERY, ids=[f"{b.replace(' ', '_')}_h{h}" for b, h, _ in _EVERY])
def test_every_backbone_at_a_saturating_batch(bb, hidden, kw):
    """every HIP backbone (the BASELINE configs' and the rest of the registry, + the two QAT cells) on 16 400 x 200 frames — thousands of
    sequence groups, every workgroup with several passes: loss and gradient of the train step are additive over a ragged split of the
    batch, the same launch twice is bit-identical, and the full-size forward agrees with the oracle on randomly drawn frames"""
    from types import SimpleNamespace
    from opendpd_amd import CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    name = bb.split()[0]
    Bn, Tn = 16400, 200
    torch.manual_seed(0)
    net = CoreModel(2, hidden, 1, name, **kw)
    bits = (0, 0)
    if " " in bb:
        from opendpd_amd.quant import get_quant_model
        net = get_quant_model(SimpleNamespace(quant=True, n_bits_w=8, n_bits_a=8, pretrained_model=""), net)
        bits = (8, 8)
    net = net.cuda().train()
    with torch.no_grad():
        for k, p in net.named_parameters():
            if k == "backbone.rru.Z":
                p.uniform_(-0.5, 0.5)
            if k == "backbone.cs":
                p.mul_(min(1.0, 1.5 / float(p.abs().sum())))
    g = torch.Generator().manual_seed(3)
    amp, ph = 0.05 + 0.85 * torch.rand(Bn, Tn, 1, generator=g), 2 * np.pi * torch.rand(Bn, Tn, 1, generator=g)
    x = torch.cat((amp * torch.cos(ph), amp * torch.sin(ph)), -1).cuda()
    t = (0.3 * torch.randn(Bn, Tn, 2, generator=g)).cuda()
    count = Bn * Tn * 2

    def grad(sl):
        opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
        loss = fused_train_step(opt, x[sl].contiguous(), t[sl].contiguous(), "l2", 0.0, global_count=count)
        return float(loss), opt.grad[:-4].clone()

    l, gr = grad(slice(0, Bn))
    cut = 9001
    la, ga = grad(slice(0, cut))
    lb, gb = grad(slice(cut, Bn))
    assert np.isfinite(l) and abs((la + lb) - l) < 5e-6 * abs(l)
    # thresholded deltas / quantisers act per sequence: the split changes summation order only
    assert rel_err((ga + gb).cpu().numpy(), gr.cpu().numpy()) < 5e-5
    l2, g2 = grad(slice(0, Bn))
    assert l2 == l and torch.equal(g2, gr)
    net.eval()
    with torch.no_grad():
        y = net(x)
    assert y.shape == (Bn, Tn, 2) and bool(torch.isfinite(y).all())
    idx = np.sort(np.random.RandomState(1).choice(Bn, 24, replace=False))
    m = make_model(name, hidden, kw.get("thx", 0), kw.get("thh", 0), bits_w=bits[0] or kw.get("num_dvr_units", 0), bits_a=bits[1])
    p = net.backbone.flat_params().detach().cpu().numpy()
    if bits[0]:               # eval mode: the 16-bit output quantiser of fc_out is on (quant_layers.py:77-80)
        yo = Oracle("f32").qat_forward(m, p, x[idx].cpu().numpy(), eval_mode=True)
    else:
        yo, _ = Oracle("f32").forward(m, p, x[idx].cpu().numpy())
    tol = 5e-3 if "delta" in name else 2e-5           # a threshold decision within rounding of its corner derails one sequence
    err = np.abs(y[idx].cpu().numpy() - yo).reshape(len(idx), -1).max(1) / np.abs(yo).max()
    assert (err < 2e-5).sum() >= len(idx) - 1 and err.max() < tol, err
