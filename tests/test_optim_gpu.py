"""The fused optimiser kinds (csrc/optim.hip, odpd_clip_optim_step) against torch.optim — the objects project.py:274-297 builds:
AdamW(lr), Adam(lr), SGD(lr, momentum=0.9), RMSprop(lr) — on the same gradients over several steps, with clip_grad_norm_ in front,
with and without a skip mask (parameters whose .grad is None)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _torch_opt(kind, params, lr):
    return {"adamw": lambda: torch.optim.AdamW(params, lr=lr), "adam": lambda: torch.optim.Adam(params, lr=lr),
            "sgd": lambda: torch.optim.SGD(params, lr=lr, momentum=0.9), "rmsprop": lambda: torch.optim.RMSprop(params, lr=lr)}[kind]()


@pytest.mark.parametrize("kind", ["adamw", "adam", "sgd", "rmsprop"])
@pytest.mark.parametrize("max_norm", [0.0, 0.5])
@pytest.mark.parametrize("masked", [False, True])
def test_fused_optimizer_kinds_match_torch(kind, max_norm, masked):
    from opendpd_amd import _lib
    lib = _lib.load()
    P, lr = 1041, 3e-3
    g = torch.Generator().manual_seed(7)
    p0 = torch.randn(P, generator=g) * 0.4
    skip = (torch.rand(P, generator=g) < 0.1) if masked else torch.zeros(P, dtype=torch.bool)
    # reference: two tensors — the stepped one and the one whose grad stays None
    ref = torch.nn.Parameter(p0.clone())
    opt = _torch_opt(kind, [ref], lr)
    flat = p0.clone().cuda()
    s1, s2 = torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda")
    norm = torch.zeros(1, device="cuda")
    sk = skip.to(torch.uint8).cuda()
    for step in range(1, 8):
        grad = torch.randn(P, generator=g) * (0.05 if step % 2 else 0.01)
        ge = grad.clone()
        ge[skip] = 0.0
        # torch: masked entries have no gradient -> emulate by restoring them after the step (they are outside the norm as well)
        ref.grad = ge.clone()
        if max_norm:
            torch.nn.utils.clip_grad_norm_([ref], max_norm)
        before = ref.detach().clone()
        opt.step()
        with torch.no_grad():
            ref[skip] = before[skip]
        gd = torch.cat([grad, torch.zeros(4)]).cuda()
        rc = lib.odpd_clip_optim_step(_lib.stream_ptr(), _lib.OPTIMIZER_IDS[kind], P, _lib.ptr(flat), _lib.ptr(gd), _lib.ptr(s1), _lib.ptr(s2),
                                      step, float(lr), float(max_norm), _lib.ptr(norm), _lib.ptr(sk) if masked else None)
        assert rc == 0
        assert abs(float(norm) - float(ge.norm())) < 1e-5 * float(ge.norm())
        err = float((flat.cpu() - ref.detach()).abs().max())
        assert err < (2e-7 if kind in ("sgd", "rmsprop") else 6e-7), (kind, step, err)       # |p| up to 1.6: one / a few ulp
    if masked:      # untouched: value and state
        assert torch.equal(flat.cpu()[skip], p0[skip]) and float(s1.cpu()[skip].abs().max()) == 0.0 and float(s2.cpu()[skip].abs().max()) == 0.0
    # Adam / AdamW with weight decay off on masked entries etc. are covered by the equality above; the state of torch's optimiser agrees too
    st = opt.state[ref]
    if kind == "sgd":
        assert float((s1.cpu()[~skip] - st["momentum_buffer"][~skip]).abs().max()) < 1e-7
    elif kind == "rmsprop":
        assert float((s2.cpu()[~skip] - st["square_avg"][~skip]).abs().max()) < 1e-9
    else:
        assert float((s1.cpu()[~skip] - st["exp_avg"][~skip]).abs().max()) < 1e-8


def test_bad_arguments():
    from opendpd_amd import _lib
    lib = _lib.load()
    t = torch.zeros(16, device="cuda")
    assert lib.odpd_clip_optim_step(_lib.stream_ptr(), 9, 12, _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), 1, 1e-3, 0.0, None, None) != 0
    assert lib.odpd_clip_optim_step(_lib.stream_ptr(), 2, 12, _lib.ptr(t), _lib.ptr(t), None, _lib.ptr(t), 1, 1e-3, 0.0, None, None) != 0
    assert lib.odpd_clip_optim_step(_lib.stream_ptr(), 3, 12, _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), 0, 1e-3, 0.0, None, None) != 0
