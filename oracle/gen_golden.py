#!/usr/bin/env python3
"""Golden-vector generator (TEST INFRASTRUCTURE — runs only in the build container).

Imports the *reference* implementation from /root/reference (read-only), runs its
CPU path on small seeded inputs and writes input/output vectors to tests/golden/*.npz.
Only the vectors are committed; the reference never travels to the GPU box.

What is pinned (reference file:line the vectors exercise):
  * backbones: gru.py:45-48, dgru.py:59-74, lstm.py:45-48, vdlstm.py:56-81,
    deltagru.py:59-77 + :211-264, deltagru_tcnskip.py:87-103 + :248-293,
    tcnn.py:82-97, pgjanet.py:26-76, qgru.py:59-71, qgru_amp1.py:59-76, gmp.py:18-50, rvtdcnn.py:35-62, neuraltx.py:116-137, deltajanet.py:50-64 + :211-274, dvrjanet.py:44-101, bojanet.py:55-106, apnrru.py:54-131, mcldnn.py:99-134
  * registry models.py:10-160 (CoreModel) and models.py:163-176 (CascadedModel)
  * train step modules/train_funcs.py:33-44 (zero_grad, fwd, MSE, bwd, clip 200, AdamW)
  * quant path quant/__init__.py:20-37 -> quant_envs.py:138-306
  * metrics utils/metrics.py:42-187, target gain utils/util.py:26-33
  * framing modules/data_collector.py:203-252

Usage:  python oracle/gen_golden.py            (writes tests/golden/)
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

# harness-side bridge for reference defect: quant/__init__ does not export Sqrt/Pow
import quant  # noqa: E402
from quant.modules.ops import Sqrt, Pow  # noqa: E402

quant.Sqrt, quant.Pow = Sqrt, Pow

import models as ref_models  # noqa: E402
from modules.data_collector import load_dataset, IQFrameDataset, IQSegmentDataset  # noqa: E402
from utils import metrics as ref_metrics  # noqa: E402
from utils.util import set_target_gain  # noqa: E402

LR, CLIP = 5e-4, 200.0


def real_frames(name, B, T, seed, split="train"):
    """B frames of length T cut from the bundled measured PA data (input, output)."""
    Xtr, ytr, Xv, yv, Xte, yte = load_dataset(dataset_name=name)
    X, y = (Xtr, ytr) if split == "train" else (Xte, yte)
    rng = np.random.RandomState(seed)
    starts = rng.randint(0, X.shape[0] - T, size=B)
    x = np.stack([X[s:s + T] for s in starts]).astype(np.float32)
    t = np.stack([y[s:s + T] for s in starts]).astype(np.float32)
    return x, t


def build(backbone, hidden, seed, thx=0.0, thh=0.0, num_layers=1, num_dvr_units=None):
    torch.manual_seed(seed)
    if backbone == "pgjanet":
        # reference defect 1: registry passes window_size= which PGJANET.__init__ lacks -> construct directly
        from backbones.pgjanet import PGJANET
        net = ref_models.CoreModel.__new__(ref_models.CoreModel)
        torch.nn.Module.__init__(net)
        net.output_size, net.input_size, net.hidden_size, net.num_layers = 2, 2, hidden, 1
        net.backbone_type = backbone
        net.backbone = PGJANET(hidden_size=hidden, output_size=2, bias=True)
        net.backbone.reset_parameters()
        return net
    return ref_models.CoreModel(input_size=2, hidden_size=hidden, num_layers=num_layers, backbone_type=backbone,
                                thx=thx, thh=thh, num_dvr_units=num_dvr_units)


def sd_np(net, prefix):
    return {f"{prefix}/{k}": v.detach().cpu().numpy().copy() for k, v in net.state_dict().items()}


def reset_stats(net):
    bb = getattr(net, "backbone", None)
    if bb is not None and hasattr(bb, "set_debug"):
        bb.set_debug(1)
    elif bb is not None and hasattr(getattr(bb, "rnn", None), "set_debug"):      # DeltaJANET: only its layer has the counters
        bb.rnn.set_debug(1)


def read_stats(net):
    bb = getattr(net, "backbone", None)
    if bb is None or not (hasattr(bb, "get_temporal_sparsity") or hasattr(getattr(bb, "rnn", None), "get_temporal_sparsity")):
        return {}
    st = bb.rnn.statistics
    return {"stats": np.array([float(st["num_dx_zeros"]), float(st["num_dx_numel"]),
                               float(st["num_dh_zeros"]), float(st["num_dh_numel"])], dtype=np.float64)}


def step_case(net, x, tgt, n_steps=3, trainable=None, loss="l2"):
    """fwd / loss / bwd / clip / AdamW exactly as modules/train_funcs.py:33-44."""
    out = {}
    xt = torch.from_numpy(x).clone().requires_grad_(True)
    tt = torch.from_numpy(tgt)
    crit = torch.nn.MSELoss() if loss == "l2" else torch.nn.L1Loss()
    params = [p for p in net.parameters()]
    opt = torch.optim.AdamW(params, lr=LR)
    net.train()
    losses = []
    for s in range(n_steps + 1):
        opt.zero_grad()
        if xt.grad is not None:
            xt.grad = None
        reset_stats(net)
        y = net(xt)
        l = crit(y, tt)
        l.backward()
        losses.append(float(l.item()))
        if s == 0:
            out["y"] = y.detach().numpy().copy()
            out["gx"] = xt.grad.detach().numpy().copy()
            out.update(read_stats(net))
            for k, p in net.named_parameters():
                if p.grad is not None:
                    out[f"g/{k}"] = p.grad.detach().numpy().copy()
        if s == n_steps:
            break
        torch.nn.utils.clip_grad_norm_(net.parameters(), CLIP)
        opt.step()
        out.update(sd_np(net, f"p{s + 1}"))
    out["losses"] = np.array(losses, dtype=np.float64)
    # optimiser state after the last step (exp_avg / exp_avg_sq) keyed by parameter name
    names = [k for k, _ in net.named_parameters()]
    for k, p in zip(names, params):
        st = opt.state.get(p, None)
        if st:
            out[f"m{n_steps}/{k}"] = st["exp_avg"].numpy().copy()
            out[f"v{n_steps}/{k}"] = st["exp_avg_sq"].numpy().copy()
    return out


def save(name, d):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KB, {len(d)} arrays)")


def gen_backbones(only=None):
    cases = [
        # name, backbone, H, thx, thh
        ("gru_h11", "gru", 11, 0, 0),
        ("gru_h23", "gru", 23, 0, 0),
        ("dgru_h13", "dgru", 13, 0, 0),
        ("dgru_h8", "dgru", 8, 0, 0),
        ("dgru_h23", "dgru", 23, 0, 0),
        ("lstm_h14", "lstm", 14, 0, 0),
        ("vdlstm_h13", "vdlstm", 13, 0, 0),
        ("deltagru_h15_dense", "deltagru", 15, 0.0, 0.0),
        ("deltagru_h15_th", "deltagru", 15, 0.01, 0.05),
        ("tres_h15_dense", "deltagru_tcnskip", 15, 0.0, 0.0),
        ("tres_h15_th", "deltagru_tcnskip", 15, 0.01, 0.05),
        ("deltagru_h24_th", "deltagru", 24, 0.01, 0.05),          # two unit tiles of the S16 mapping
        ("tres_h30_th", "deltagru_tcnskip", 30, 0.005, 0.02),
        ("tcnn_c35", "tcnn", 35, 0, 0),
        ("pgjanet_h11", "pgjanet", 11, 0, 0),
        ("qgru_h10", "qgru", 10, 0, 0),
        ("qgru_h16", "qgru", 16, 0, 0),
        ("qgru_amp1_h10", "qgru_amp1", 10, 0, 0),
        ("gmp_m11", "gmp", 11, 0, 0),                             # models.py:26-28: GMP() — memory 11, degree 5 whatever hidden_size is
        ("rvtdcnn_h25", "rvtdcnn", 25, 0, 0),                     # models.py:80-81: fc_hid_size = hidden_size; 1007 parameters
        ("rvtdcnn_h6", "rvtdcnn", 6, 0, 0),                       # the reference's own default fc_hid_size (rvtdcnn.py:11)
        ("neuraltx_c36", "neuraltx", 36, 0, 0),                   # 986 parameters
        ("neuraltx_c12", "neuraltx", 12, 0, 0),
        ("deltajanet_h15", "deltajanet", 15, 0.01, 0.05),         # the wrapper drops the thresholds (deltajanet.py:23-27): dense deltas
        ("deltajanet_h22", "deltajanet", 22, 0, 0),               # 1366 parameters; two unit tiles of the S16 mapping
        ("dvrjanet_h12_k3", "dvrjanet", 12, 0, 0),                # 1097 parameters at the CLI's default num_dvr_units = 3
        ("dvrjanet_h8_k4", "dvrjanet", 8, 0, 0),
        ("bojanet_h12", "bojanet", 12, 0, 0),                     # 818 parameters; phase re-rotation wraps once (units 6..11 reuse filters 0..5)
        ("bojanet_h16", "bojanet", 16, 0, 0),                     # wraps twice
        ("bojanet_h5", "bojanet", 5, 0, 0),                       # fewer units than filters
        ("apnrru_h8", "apnrru", 8, 0, 0),                         # 903 parameters; state of 2 x 8 + 3 = 19
        ("apnrru_h14", "apnrru", 14, 0, 0),                       # state of 31: the kernels' largest
        ("apnrru_h5", "apnrru", 5, 0, 0),
        ("mcldnn_c8", "mcldnn", 8, 0, 0),                         # 2109 parameters
        ("mcldnn_c3", "mcldnn", 3, 0, 0),
    ]
    dvr_units = {"dvrjanet_h12_k3": 3, "dvrjanet_h8_k4": 4}
    x, tgt = real_frames("DPA_200MHz", 5, 37, seed=1)       # ragged: B%4!=0, odd T
    xa, ta = real_frames("APA_200MHz", 8, 200, seed=2)      # config-shaped frames (T=200)
    for name, bb, H, thx, thh in cases:
        if only and name not in only:
            continue
        net = build(bb, H, seed=0, thx=thx, thh=thh, num_dvr_units=dvr_units.get(name))
        if bb == "dvrjanet":      # xavier weights and zero biases leave the DVR knots (k/K) far from W_ax |x| + W_ah hs: move some biases
            with torch.no_grad():  # and scale the input columns so that every knot sees both signs
                g = torch.Generator().manual_seed(5)
                for k, p in net.named_parameters():
                    if k.endswith("bias"):
                        p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 0.4)
                net.backbone.W_ax.weight.mul_(1.5)
        if bb == "apnrru":        # Z = 0 at construction switches the deep cell off (s' = sigmoid(C s)): give it, and the biases, values
            with torch.no_grad():
                g = torch.Generator().manual_seed(7)
                net.backbone.rru.Z.copy_((torch.rand(net.backbone.rru.Z.shape, generator=g) - 0.5) * 1.2)
                for k, p in net.named_parameters():
                    if k.endswith("bias"):
                        p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 0.4)
        if bb == "mcldnn":        # zero biases at construction
            with torch.no_grad():
                g = torch.Generator().manual_seed(8)
                for k, p in net.named_parameters():
                    if "bias" in k:
                        p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 0.4)
        if bb == "bojanet":       # zero biases at construction: give the bias gradients something to be checked against
            with torch.no_grad():
                g = torch.Generator().manual_seed(6)
                for k, p in net.named_parameters():
                    if k.endswith("bias"):
                        p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 0.4)
        d = {"x": x, "tgt": tgt, "meta": np.array(json.dumps(
            {"backbone": bb, "hidden": H, "thx": thx, "thh": thh, "lr": LR, "clip": CLIP, "num_dvr_units": dvr_units.get(name, 0),
             "n_param": int(sum(p.numel() for p in net.parameters()))}))}
        d.update(sd_np(net, "sd"))
        # config-shaped forward + loss only (before any parameter update)
        net.eval()
        with torch.no_grad():
            reset_stats(net)
            ya = net(torch.from_numpy(xa))
            d["xa"], d["ta"], d["ya"] = xa, ta, ya.numpy().copy()
            d["loss_a"] = np.array(float(torch.nn.functional.mse_loss(ya, torch.from_numpy(ta))))
            sa = read_stats(net)
            if sa:
                d["stats_a"] = sa["stats"]
        d.update(step_case(net, x, tgt))
        save(name, d)


# models that moved onto kernels in r04 (csrc/gru_wide.hip, lstm_wide.hip: one layer of 33 .. 64 units; gru_layers2.hip: two layers; `wide_more`): same recipe as gen_wide
WIDE_MORE = [("wide_gru_h48", "gru", 48, 1, 0, 0), ("wide_dgru_h64", "dgru", 64, 1, 0, 0), ("wide_lstm_h40", "lstm", 40, 1, 0, 0),
             ("wide_dgru_h13_l2", "dgru", 13, 2, 0, 0)]


def gen_wide(cases=None):
    """Configurations beyond the HIP kernels' envelope (two layers / hidden > 32): pins backbones/wide.py to the reference."""
    cases = cases or [("wide_gru_h12_l2", "gru", 12, 2, 0, 0), ("wide_dgru_h40", "dgru", 40, 1, 0, 0), ("wide_lstm_h10_l2", "lstm", 10, 2, 0, 0),
             ("wide_vdlstm_h36", "vdlstm", 36, 1, 0, 0), ("wide_qgru_amp1_h34", "qgru_amp1", 34, 1, 0, 0),
             ("wide_deltagru_h34", "deltagru", 34, 1, 0.01, 0.05), ("wide_tres_h33", "deltagru_tcnskip", 33, 1, 0.01, 0.05),
             ("wide_pgjanet_h18", "pgjanet", 18, 1, 0, 0), ("wide_tcnn_c66", "tcnn", 66, 1, 0, 0)]
    x, tgt = real_frames("DPA_200MHz", 3, 21, seed=5)
    for name, bb, H, L, thx, thh in cases:
        net = build(bb, H, seed=0, thx=thx, thh=thh, num_layers=L)
        d = {"x": x, "tgt": tgt, "meta": np.array(json.dumps(
            {"backbone": bb, "hidden": H, "num_layers": L, "thx": thx, "thh": thh, "lr": LR, "clip": CLIP,
             "n_param": int(sum(p.numel() for p in net.parameters()))}))}
        d.update(sd_np(net, "sd"))
        d.update(step_case(net, x, tgt, n_steps=1))
        save(name, d)


def gen_cascade():
    """train_dpd: DPD (TRes-DeltaGRU H15 / DGRU H13) -> frozen PA DGRU H23 (steps/train_dpd.py:28-63)."""
    x, _ = real_frames("APA_200MHz", 5, 37, seed=3)
    tgt = (1.0 * x).astype(np.float32)  # target_gain(APA_200MHz) == 1.0 -> targets = g * X
    for name, bb, H, thx, thh in [("cascade_tres15_dgru23", "deltagru_tcnskip", 15, 0.01, 0.05),
                                  ("cascade_dgru13_dgru23", "dgru", 13, 0, 0),
                                  ("cascade_gru11_gru11", "gru", 11, 0, 0)]:
        pa_bb, pa_h = ("gru", 11) if "gru11_gru11" in name else ("dgru", 23)
        pa = build(pa_bb, pa_h, seed=7)
        dpd = build(bb, H, seed=0, thx=thx, thh=thh)
        net = ref_models.CascadedModel(dpd_model=dpd, pa_model=pa)
        net.freeze_pa_model()
        d = {"x": x, "tgt": tgt, "meta": np.array(json.dumps(
            {"dpd": bb, "dpd_hidden": H, "pa": pa_bb, "pa_hidden": pa_h, "thx": thx, "thh": thh,
             "lr": LR, "clip": CLIP}))}
        d.update(sd_np(net, "sd"))
        d.update(step_case(net, x, tgt))
        assert all(p.grad is None for p in net.pa_model.parameters())
        save(name, d)


def gen_quant():
    """QAT qgru (quant/__init__.py:20-37): W8A8 and W16A16; train- and eval-mode outputs, STE grads."""
    class P:  # the attributes get_quant_model reads from the Project object
        quant = True
        pretrained_model = ""
        quant_dir_label = ""
    x, tgt = real_frames("DPA_200MHz", 5, 37, seed=4)
    for bits in (8, 16):
        for bb, H in (("qgru", 10), ("qgru_amp1", 10)):
            torch.manual_seed(0)
            fnet = build(bb, H, seed=0)
            P.n_bits_w = P.n_bits_a = bits
            torch.manual_seed(123)  # _reset_pygru consumes RNG
            qnet = quant.get_quant_model(P, fnet)
            assert qnet is not fnet, "quantisation fell back to the float model"
            d = {"x": x, "tgt": tgt, "meta": np.array(json.dumps(
                {"backbone": bb, "hidden": H, "bits": bits, "lr": LR, "clip": CLIP,
                 "n_param": int(sum(p.numel() for p in qnet.parameters()))}))}
            d.update(sd_np(qnet, "sd"))
            qnet.eval()
            with torch.no_grad():
                d["y_eval"] = qnet(torch.from_numpy(x)).numpy().copy()
            d.update(step_case(qnet, x, tgt))
            # forward with the parameters reached after the three steps (non-zero biases, decayed scales)
            qnet.train()
            with torch.no_grad():
                d["y_p3_train"] = qnet(torch.from_numpy(x)).numpy().copy()
            qnet.eval()
            with torch.no_grad():
                d["y_p3_eval"] = qnet(torch.from_numpy(x)).numpy().copy()
            d.update(sd_np(qnet, "sd3"))   # full state (params + buffers) after the steps
            save(f"quant_{bb}_h{H}_w{bits}a{bits}", d)


def gen_metrics_and_framing():
    """Known answers for NMSE/EVM/ACLR on raw bundled data + framing/segment shapes + target gain."""
    res = {}
    for ds in ("DPA_200MHz", "APA_200MHz", "APA_200MHz_b"):
        spec = json.load(open(os.path.join(REF, "datasets", ds, "spec.json")))
        Xtr, ytr, Xv, yv, Xte, yte = load_dataset(dataset_name=ds)
        g = float(set_target_gain(Xtr, ytr))
        seg = IQSegmentDataset(Xte, yte, nperseg=spec["nperseg"])
        pred = seg.targets.numpy()           # "prediction" = measured PA output segments
        truth = IQSegmentDataset(Xte, g * Xte, nperseg=spec["nperseg"]).targets.numpy()
        nmse = float(ref_metrics.NMSE(pred, truth))
        evm = float(ref_metrics.EVM(pred, truth, bw_main_ch=spec["bw_main_ch"], n_sub_ch=spec["n_sub_ch"],
                                    nperseg=spec["nperseg"]))
        al, ar = ref_metrics.ACLR(pred, fs=spec["input_signal_fs"], nperseg=spec["nperseg"],
                                  bw_main_ch=spec["bw_main_ch"], n_sub_ch=spec["n_sub_ch"])
        res[ds] = {"target_gain": g, "NMSE": nmse, "EVM": evm, "ACLR_L": float(al), "ACLR_R": float(ar),
                   "n_train": int(Xtr.shape[0]), "n_test": int(Xte.shape[0]), "seg_shape": list(pred.shape)}
    # small self-contained metric vector: 2 segments of 512 synthetic samples, fs/bw made up
    rng = np.random.RandomState(5)
    pred = rng.randn(2, 512, 2).astype(np.float32) * 0.3
    truth = (pred + 0.05 * rng.randn(2, 512, 2)).astype(np.float32)
    small = {"pred": pred, "truth": truth,
             "NMSE": np.array(ref_metrics.NMSE(pred, truth)),
             "EVM": np.array(ref_metrics.EVM(pred, truth, bw_main_ch=200e6, n_sub_ch=2, nperseg=512)),
             "ACLR": np.array(ref_metrics.ACLR(pred, fs=800e6, nperseg=512, bw_main_ch=200e6, n_sub_ch=2))}
    # framing (data_collector.py:233-252): stream of 300 samples, F=50, stride 1 and 7
    stream = rng.randn(300, 2)
    for s in (1, 7):
        fs = IQFrameDataset(stream, 2 * stream, frame_length=50, stride=s)
        small[f"frames_s{s}_shape"] = np.array(fs.features.shape)
        small[f"frames_s{s}_first"] = fs.features[0].numpy()
        small[f"frames_s{s}_last"] = fs.features[-1].numpy()
    small["stream"] = stream
    segs = IQSegmentDataset(stream, 2 * stream, nperseg=128)
    small["segs"] = segs.features.numpy()
    # DataLoader shuffle order for seed 0 (project.py:108-112,236): first 16 indices of a 22991-frame epoch
    torch.manual_seed(0)
    small["perm_head_22991"] = torch.randperm(22991)[:16].numpy()
    save("metrics_framing", small)
    with open(os.path.join(OUT, "dataset_known_answers.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    which = sys.argv[1:] or ["backbones", "cascade", "quant", "metrics"]
    only = [w.split("=", 1)[1].split(",") for w in which if w.startswith("only=")]      # e.g. backbones only=gru_h11,dgru_h8
    if "backbones" in which:
        gen_backbones(only[0] if only else None)
    if "wide" in which:
        gen_wide()
    if "wide_more" in which:
        gen_wide(WIDE_MORE)
    if "cascade" in which:
        gen_cascade()
    if "quant" in which:
        gen_quant()
    if "metrics" in which:
        gen_metrics_and_framing()
