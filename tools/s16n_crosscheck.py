#!/usr/bin/env python3
"""Every launch flavour of the hidden-17..32 GRU-family kernels (csrc/gru_s16n.hip) against the C oracle, forced onto the S16 mapping
at small batches: the frozen-PA single-launch step of a cascade (forward + loss + dL/dx), the frozen backward (dL/dx only), the
weight-gradient backward, both together, and the fused train step — all four feature sets, hidden 17..32, ragged shapes.
`check(pbb, ph)` is what tests/test_sweeps_gpu.py runs per (backbone, hidden size); as a script it prints the whole table.
usage (GPU box): PYTHONPATH=. python tools/s16n_crosscheck.py"""
import warnings

import numpy as np
import torch

SHAPES = ((1, 5), (3, 32), (16, 33), (33, 65), (70, 9), (200, 5), (16 * 300 + 3, 4))     # incl. 5..301 sequence groups (several loss rows)
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def check(pbb, ph, shapes=SHAPES, seed=3):
    """-> (worst error per flavour over the well-conditioned draws, cases beyond tolerance, draws on a kink, number of cases).  Expects the
    S16 mapping forced (odpd_set_tuning("s16_min_batch", 0))."""
    from opendpd_amd import CascadedModel, CoreModel
    from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
    from oracle.oracle import Oracle, make_model
    o, o64 = Oracle("f32"), Oracle("f64")
    rng = np.random.RandomState(seed + 97 * ph)
    worst, bad, kinks, n = [0.0] * 5, [], [], 0
    for B, T in shapes:
        torch.manual_seed(B * 100 + T + ph)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            dpd, pa = CoreModel(2, 6, 1, "gru"), CoreModel(2, ph, 1, pbb)
        with torch.no_grad():
            for k, p in pa.named_parameters():
                if "bias" in k:
                    p.uniform_(-0.2, 0.2)
        net = CascadedModel(dpd_model=dpd, pa_model=pa)
        net.freeze_pa_model()
        net = net.cuda()
        md, mp = make_model("gru", 6), make_model(pbb, ph)
        pd = dpd.backbone.flat_params().detach().cpu().numpy().copy()
        pp = pa.backbone.flat_params().detach().cpu().numpy().copy()
        x = (rng.uniform(0.1, 0.8, (B, T, 2)) * rng.choice([-1.0, 1.0], (B, T, 2))).astype(np.float32)
        t = (0.5 * rng.randn(B, T, 2)).astype(np.float32)
        u, _ = o.forward(md, pd, x)
        y, _ = o.forward(mp, pp, u)
        lo, dy = o.loss("l2", y, t)
        gp, du = o.backward(mp, pp, u, dy)
        gd, _ = o.backward(md, pd, x, du, need_dx=False)
        # conditioning of the draw: the PA's polar features divide by |u|; a DPD output next to the origin amplifies rounding, and
        # the fp32 oracle then leaves the fp64 one by as much as the kernels leave the fp32 one
        f8 = lambda v: np.asarray(v, dtype=np.float64)
        y8, _ = o64.forward(mp, f8(pp), f8(u))
        l8, dy8 = o64.loss("l2", y8, f8(t))
        gp8, du8 = o64.backward(mp, f8(pp), f8(u), f8(dy))
        cond = max(rel(y, y8), rel(du, du8), rel(gp, gp8), abs(lo - l8) / abs(l8))
        e = [0.0] * 5
        # 1. cascade step (frozen PA: forward + loss + dL/dx in one launch)
        opt = FusedAdamW(net, lr=0.0, weight_decay=0.0)
        lg = float(fused_train_step(opt, torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(), "l2", 0.0))
        e[0] = max(abs(lg - lo) / abs(lo) * 10, rel(opt.grad[:-4].cpu().numpy(), gd))
        # 2. frozen backward through autograd (dL/dx only)
        ut = torch.from_numpy(u).cuda().requires_grad_(True)
        yg = pa(ut)
        yg.backward(torch.from_numpy(dy).cuda())
        e[1] = max(rel(ut.grad.cpu().numpy(), du), 10 * rel(yg.detach().cpu().numpy(), y))
        # 3. / 4. weight gradients alone, and with dL/dx
        for q in pa.parameters():
            q.requires_grad_(True)
        pa(torch.from_numpy(u).cuda()).backward(torch.from_numpy(dy).cuda())
        e[2] = rel(np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in pa.parameters()]), gp)
        for q in pa.parameters():
            q.grad = None
        ut = torch.from_numpy(u).cuda().requires_grad_(True)
        pa(ut).backward(torch.from_numpy(dy).cuda())
        e[3] = max(rel(np.concatenate([q.grad.cpu().numpy().reshape(-1) for q in pa.parameters()]), gp), rel(ut.grad.cpu().numpy(), du))
        # 5. fused train step of the model on its own
        opt2 = FusedAdamW(pa, lr=0.0, weight_decay=0.0)
        l2 = float(fused_train_step(opt2, torch.from_numpy(u).cuda(), torch.from_numpy(t).cuda(), "l2", 0.0))
        e[4] = max(abs(l2 - lo) / abs(lo) * 10, rel(opt2.grad[:-4].cpu().numpy(), gp))
        n += 1
        if max(e) > 3e-4 + 100 * cond or not np.isfinite(e).all():
            # a relu kink of the DGRU head within rounding of 0 makes the gradient itself discontinuous (one sequence jumps): then
            # the fp64 ORACLE's gradients move by as much under a 2e-6 relative change of u, and the case says nothing about the kernels
            def grads64(scale):
                y9, _ = o64.forward(mp, f8(pp), f8(u) * scale)
                _, dy9 = o64.loss("l2", y9, f8(t))
                return o64.backward(mp, f8(pp), f8(u) * scale, dy9)
            (ga, da), (gb, db), (gc, dc) = grads64(1.0), grads64(1 + 2e-6), grads64(1 - 2e-6)
            jump = max(rel(db, da), rel(dc, da), rel(gb, ga), rel(gc, ga))
            umin = float(np.sqrt((u ** 2).sum(-1)).min())
            if jump > 0.3 * max(e) or umin < 2e-3:
                kinks.append((pbb, ph, B, T, "max err %.1e; fp64 oracle under a 2e-6 input change: %.1e; min |u| %.1e" % (max(e), jump, umin)))
                continue
            bad.append((pbb, ph, B, T, ["%.1e" % v for v in e], "oracle fp32 vs fp64 %.1e, min |u| %.1e" % (cond, umin)))
        elif cond <= 3e-6:
            worst = [max(a, b) for a, b in zip(worst, e)]
    return worst, bad, kinks, n


if __name__ == "__main__":
    from opendpd_amd import _lib
    lib = _lib.load()
    lib.odpd_set_tuning(b"s16_min_batch", 0)
    bad, kinks, n = [], [], 0
    for pbb in ("dgru", "gru", "qgru", "qgru_amp1"):
        worst = [0.0] * 5
        for ph in range(17, 33):
            w, b, k, m = check(pbb, ph)
            worst = [max(a, c) for a, c in zip(worst, w)]
            bad += b
            kinks += k
            n += m
        print(f"{pbb:10s} worst  cascade step {worst[0]:.2e}  frozen bwd {worst[1]:.2e}  wgrad {worst[2]:.2e}  wgrad+dx {worst[3]:.2e}  fused train {worst[4]:.2e}", flush=True)
    lib.odpd_set_tuning(b"s16_min_batch", -1)
    print(f"{n} cases, {len(bad)} beyond tolerance")
    print(f"{len(kinks)} draw(s) on a kink / next to the origin: {kinks}")
    for b in bad[:40]:
        print("  ", b)
