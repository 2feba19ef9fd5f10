"""Fused train step (odpd_train_fwd_bwd: forward + loss + BPTT with weight gradients, modules/train_funcs.py:28-48) of the GRU-family models
with 17 .. 24 hidden units — train_pa of the reference's default PA, dgru H23 (bash_scripts/OpenDPDv2.sh:39-52, backbones/dgru.py:59-74) — on
the bf16 matrix pipe with three-way operand splits (csrc/gru_s16x.hip, gru16x_train_kernel, r06).  The PARAMETER gradients are checked against
the fp64 oracle at fp32-level tolerances and against the exact-fp32 kernel the step ran on before (csrc/gru_s16n.hip, knob "s16x_train" = 0),
on ragged shapes (partial 16-sequence groups, odd T = a one-step tail block, T below / across the staged chunk, several workgroups)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def s16_lib():
    from opendpd_amd import _lib
    lib = _lib.load()
    assert lib.odpd_set_tuning(b"s16_min_batch", 0) == 0
    yield lib
    lib.odpd_set_tuning(b"s16_min_batch", -1)
    lib.odpd_set_tuning(b"s16x_train", 1)


def _train(lib, bb, x, t, loss, split):
    """one odpd_train_fwd_bwd call through the raw C ABI -> (loss mean, gradient vector of P floats in fp64 = the rows summed)"""
    from opendpd_amd import _lib
    assert lib.odpd_set_tuning(b"s16x_train", split) == 0
    B, T = x.shape[:2]
    P = bb.n_flat
    rows = int(lib.odpd_partial_rows(C.byref(bb.desc), B, T, 1))
    assert rows > 0
    part = torch.full((rows, P + _lib.LOSS_COLS), float("nan"), device="cuda")
    ws = torch.empty(int(lib.odpd_train_workspace_floats(C.byref(bb.desc), B, T)), device="cuda")
    _lib.check(lib.odpd_train_fwd_bwd(_lib.stream_ptr(), C.byref(bb.desc), _lib.LOSS_IDS[loss], B, T, B * T * 2, _lib.ptr(bb.flat_params()),
                                      _lib.ptr(x), _lib.ptr(t), _lib.ptr(part), _lib.ptr(ws)), "odpd_train_fwd_bwd")
    torch.cuda.synchronize()
    assert torch.isfinite(part).all()
    s = part.double().sum(0)
    return float(s[P]) / (B * T * 2), s[:P]


def _data(B, T, seed):
    rng = np.random.RandomState(seed)
    x = (rng.uniform(0.1, 0.8, (B, T, 2)) * rng.choice([-1.0, 1.0], (B, T, 2))).astype(np.float32)
    t = (0.5 * rng.randn(B, T, 2)).astype(np.float32)
    return x, t


def _oracle_grad(bb_name, H, params, x, t, loss):
    from oracle.oracle import Oracle, make_model
    o = Oracle("f64")
    m = make_model(bb_name, H)
    pp = params.astype(np.float64)
    y, _ = o.forward(m, pp, x.astype(np.float64))
    lo, dy = o.loss(loss, y, t.astype(np.float64))
    g, _ = o.backward(m, pp, x.astype(np.float64), dy, need_dx=False)
    return lo, np.asarray(g)


@pytest.mark.parametrize("bb,H", [("dgru", 23), ("gru", 23), ("dgru", 17), ("dgru", 24), ("gru", 24), ("qgru", 20), ("qgru_amp1", 21), ("gru", 18)])
@pytest.mark.parametrize("B,T,loss", [(37, 70, "l2"), (16 * 9 + 5, 21, "l2"), (33, 201, "l1"), (5, 1, "l2"), (16, 3, "l1"), (64, 2, "l2")])
def test_split_train_kernel_parameter_gradients_against_fp64_oracle_and_exact_kernel(s16_lib, bb, H, B, T, loss):
    from opendpd_amd import CoreModel
    torch.manual_seed(H * 7 + B)
    net = CoreModel(2, H, 1, bb).cuda().backbone
    x, t = _data(B, T, H + B)
    xg, tg = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    lx, gx = _train(s16_lib, net, xg, tg, loss, 1)
    ln, gn = _train(s16_lib, net, xg, tg, loss, 0)
    lo, go = _oracle_grad(bb, H, net.flat_params().detach().cpu().numpy(), x, t, loss)
    go = torch.from_numpy(go).cuda()
    scale = float(go.abs().max())
    ex, en = float((gx - go).abs().max()) / scale, float((gn - go).abs().max()) / scale
    # fp32-equivalent: the bounds of tests/test_gru_s16x_gpu.py (the frozen step's dL/du), here for every PARAMETER gradient
    assert abs(lx - lo) < 4e-7 * max(1.0, lo), (lx, lo)
    assert ex < 1.5e-6, (ex, en)
    assert ex < 3.0 * en + 1e-7, (ex, en)
    assert float((gx - gn).abs().max()) / scale < 2e-6
    # per parameter tensor, relative to that tensor's own largest gradient (a small tensor must not hide behind a large one)
    L = [(k, p.numel()) for k, p in net.named_parameters()]
    o = 0
    for k, nel in L:
        a, b = gx[o:o + nel], go[o:o + nel]
        sc = float(b.abs().max())
        if sc > 0:
            assert float((a - b).abs().max()) / sc < 6e-6, (k, float((a - b).abs().max()) / sc)
        o += nel


def test_split_train_kernel_is_taken_for_hidden_17_to_24_only(s16_lib):
    """hidden 25 .. 32 and <= 16 keep their exact-fp32 train kernels: the knob must not change their results by a single bit."""
    from opendpd_amd import CoreModel
    for bb, H, same in [("dgru", 23, False), ("dgru", 25, True), ("gru", 32, True), ("dgru", 13, True)]:
        torch.manual_seed(H)
        net = CoreModel(2, H, 1, bb).cuda().backbone
        x, t = _data(40, 24, H)
        xg, tg = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
        _, g1 = _train(s16_lib, net, xg, tg, "l2", 1)
        _, g0 = _train(s16_lib, net, xg, tg, "l2", 0)
        assert torch.equal(g1, g0) == same, (bb, H)


def test_split_train_kernel_full_size_additivity_and_oracle_group(s16_lib):
    """8 192 x 200 (the bench shapes are 32 768 / 65 536 x 200; this one stays in seconds): the gradient of the whole batch equals the sum of
    its two halves' (several groups per wave, many workgroups, the row reduction), and a 16-sequence launch drawn from it matches the oracle."""
    from opendpd_amd import CoreModel
    torch.manual_seed(5)
    net = CoreModel(2, 23, 1, "dgru").cuda().backbone
    B, T = 8192, 200
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.rand(B, T, 2, device="cuda", generator=g) - 0.5) * 1.2 + 0.05
    t = torch.rand(B, T, 2, device="cuda", generator=g) - 0.5
    lw, gw = _train(s16_lib, net, x, t, "l2", 1)
    la, ga = _train(s16_lib, net, x[: B // 2].contiguous(), t[: B // 2].contiguous(), "l2", 1)
    lb, gb = _train(s16_lib, net, x[B // 2:].contiguous(), t[B // 2:].contiguous(), "l2", 1)
    assert abs(lw - 0.5 * (la + lb)) < 1e-6 * lw
    # (the halves were normalised by half the count; fp32 rows summed in fp64: the grouping of sequences into rows differs, hence 1e-6)
    assert float((gw - 0.5 * (ga + gb)).abs().max()) / float(gw.abs().max()) < 1e-6
    sel = slice(4096 + 16 * 7, 4096 + 16 * 8)
    xs, ts_ = x[sel].contiguous(), t[sel].contiguous()
    l1, g1 = _train(s16_lib, net, xs, ts_, "l2", 1)
    lo, go = _oracle_grad("dgru", 23, net.flat_params().detach().cpu().numpy(), xs.cpu().numpy(), ts_.cpu().numpy(), "l2")
    go = torch.from_numpy(go).cuda()
    assert abs(l1 - lo) < 4e-7 * max(1.0, lo)
    assert float((g1 - go).abs().max()) / float(go.abs().max()) < 1.5e-6


def test_split_train_kernel_on_framed_bf16_and_fp32_streams(s16_lib):
    """the framed entry point (frames addressed in place in the resident stream, fp32 or bf16 sample storage) runs the same kernel: a step on
    frames of a stream equals the step on the materialised frames bit for bit"""
    from opendpd_amd import CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, FrameBatch, fused_train_step
    T, B = 50, 96
    g = torch.Generator(device="cuda").manual_seed(3)
    xs = (torch.rand(B + T - 1, 2, device="cuda", generator=g) - 0.5) * 1.4
    ys = torch.rand(B + T - 1, 2, device="cuda", generator=g) - 0.5
    for dt in (torch.float32, torch.bfloat16):
        xq, yq = xs.to(dt), ys.to(dt)
        res = []
        for framed in (True, False):
            torch.manual_seed(11)
            net = CoreModel(2, 23, 1, "dgru").cuda()
            opt = FusedAdamW(net, lr=1e-3)
            if framed:
                x, t = FrameBatch(xq, yq, torch.arange(B, device="cuda", dtype=torch.int64), T, 1), None
            else:
                x = xq.float().unfold(0, T, 1).permute(0, 2, 1).contiguous()
                t = yq.float().unfold(0, T, 1).permute(0, 2, 1).contiguous()
            loss = fused_train_step(opt, x, t, "l2", 200.0, B * T * 2)
            res.append((float(loss.item()), net.backbone.flat_params().detach().clone()))
        assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1]), dt


def _split(lib, bb, x, dy, knob, want_w, want_dx):
    """odpd_backbone_fwd (with checkpoints) + odpd_backbone_bwd (from dL/dy) through the raw C ABI -> (y, parameter gradient | None, dL/dx | None)"""
    from opendpd_amd import _lib
    assert lib.odpd_set_tuning(b"s16x_train", knob) == 0
    B, T = x.shape[:2]
    P = bb.n_flat
    y = torch.full_like(x, float("nan"))
    ck = torch.empty(int(lib.odpd_ckpt_floats(C.byref(bb.desc), B, T)), device="cuda")
    _lib.check(lib.odpd_backbone_fwd(_lib.stream_ptr(), C.byref(bb.desc), B, T, _lib.ptr(bb.flat_params()), _lib.ptr(x), _lib.ptr(y), _lib.ptr(ck), None), "fwd")
    rows = int(lib.odpd_partial_rows(C.byref(bb.desc), B, T, 0))
    part = torch.full((rows, P + _lib.LOSS_COLS), float("nan"), device="cuda") if want_w else None
    dx = torch.full_like(x, float("nan")) if want_dx else None
    _lib.check(lib.odpd_backbone_bwd(_lib.stream_ptr(), C.byref(bb.desc), B, T, _lib.ptr(bb.flat_params()), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(ck),
                                     _lib.ptr(part) if want_w else None, _lib.ptr(dx) if want_dx else None), "bwd")
    torch.cuda.synchronize()
    return y, (part.double().sum(0)[:P] if want_w else None), dx


@pytest.mark.parametrize("bb,H", [("dgru", 23), ("gru", 23), ("dgru", 17), ("qgru", 20), ("qgru_amp1", 24)])
@pytest.mark.parametrize("B,T", [(37, 70), (16 * 9 + 5, 21), (33, 201), (5, 1)])
@pytest.mark.parametrize("want_w,want_dx", [(True, False), (True, True), (False, True)])
def test_split_forward_and_backward_against_fp64_oracle_and_exact_kernels(s16_lib, bb, H, B, T, want_w, want_dx):
    """the split entry points of hidden 17 .. 24 (r06: gru16x_fwd_kernel, gru16x_bwd_kernel<NW, DX>): forward output, parameter gradients and dL/dx
    from a given dL/dy against the fp64 oracle at the fused step's bounds, and against the exact-fp32 kernels they replace"""
    from opendpd_amd import CoreModel
    from oracle.oracle import Oracle, make_model
    torch.manual_seed(H * 5 + B)
    net = CoreModel(2, H, 1, bb).cuda().backbone
    x, _ = _data(B, T, H + B + 1)
    rng = np.random.RandomState(B + T)
    dy = (rng.randn(B, T, 2) / (B * T)).astype(np.float32)
    xg, dyg = torch.from_numpy(x).cuda(), torch.from_numpy(dy).cuda()
    yx, gx, dxx = _split(s16_lib, net, xg, dyg, 1, want_w, want_dx)
    yn, gn, dxn = _split(s16_lib, net, xg, dyg, 0, want_w, want_dx)
    o = Oracle("f64")
    m = make_model(bb, H)
    pp = net.flat_params().detach().cpu().numpy().astype(np.float64)
    yo, _ = o.forward(m, pp, x.astype(np.float64))
    go, dxo = o.backward(m, pp, x.astype(np.float64), dy.astype(np.float64), need_dx=want_dx)
    yo = torch.from_numpy(np.asarray(yo)).cuda()
    sy = float(yo.abs().max())
    assert float((yx.double() - yo).abs().max()) / sy < 1.5e-6
    assert float((yx - yn).abs().max()) / sy < 2e-6
    if want_w:
        go = torch.from_numpy(np.asarray(go)).cuda()
        sc = float(go.abs().max())
        ex, en = float((gx - go).abs().max()) / sc, float((gn - go).abs().max()) / sc
        assert ex < 1.5e-6 and ex < 3.0 * en + 1e-7, (ex, en)
    if want_dx:
        do = torch.from_numpy(np.asarray(dxo)).cuda()
        sc = float(do.abs().max())
        ex, en = float((dxx.double() - do).abs().max()) / sc, float((dxn.double() - do).abs().max()) / sc
        assert torch.isfinite(dxx).all()
        assert ex < 1.5e-6 and ex < 3.0 * en + 1e-7, (ex, en)
