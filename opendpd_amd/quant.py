"""Quantisation-aware training: drop-in for the reference's `quant.get_quant_model` (quant/__init__.py:20-37).

The reference performs model surgery (quant/quant_envs.py:138-306): nn.GRU -> Python GRU of GRUCells (re-initialised),
nn.Linear -> INT_Linear, Sigmoid/Tanh/Add/Mul modules -> Quant_* with one power-of-two scale parameter each.  Here the result of
that surgery is a single HIP-backed backbone — `QuantGRUCellModel` for gru / dgru / qgru / qgru_amp1, `QuantTResDeltaGRU` for
deltagru_tcnskip — with the same parameter / buffer names (`backbone.rnn.rnn_cell_list.0.{x2h,h2h}.{weight,bias,
weight_quantizer.scale,...}`, `...{sigmoid,tanh,add,mul}.quantizer.scale`, `backbone.fc_out.*`, ...) so state dicts are
interchangeable, and the same construction-time RNG consumption.
Kernels: csrc/qat_s16.hip (16 sequences per wave, every kind, hidden <= 32) and csrc/qgru_family.hip (qgru / qgru_amp1 at hidden <= 16
and small batches); integer-grid arithmetic bit-exact with the reference for 8-bit grids."""
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .backbones.native import NativeBackbone, init_gatewise
from .models import CoreModel


class _QScale(nn.Module):
    """INT_Quantizer state: parameter `scale`, buffers pow2_scale / decimal_num / integer_num (quantizers.py:15-48)."""

    def __init__(self, bits, init_scale):
        super().__init__()
        self.bits = bits
        self.scale = nn.Parameter(torch.Tensor([init_scale]))
        self.register_buffer("pow2_scale", torch.Tensor([0.0]))
        self.register_buffer("decimal_num", torch.Tensor([1.0]))
        self.register_buffer("integer_num", torch.Tensor([bits - 1 - 1.0]))
        self.exercised = False          # has a forward gone through this quantiser (see QuantQGRU.sync_mode)

    def quantise(self, x):
        """INT_Quantizer.forward through ATen (quantizers.py:73-92; all_positive=False, as recur_rpls_layers builds every layer
        quantiser, quant_envs.py:57-58): power-of-two scale, clamp, straight-through round.  Used by the backbones whose quantised form has no HIP
        kernels (the announced ATen route of get_quant_model); the kernel-backed models carry the same arithmetic in csrc/odpd_qat.h."""
        s = self.scale.abs()
        l = s.log2().round()
        pow2 = 2 ** l
        dec = l.abs().int()
        if dec != self.decimal_num:
            self.pow2_scale.copy_(pow2.detach()); self.decimal_num.copy_(dec); self.integer_num.copy_(self.bits - 1 - dec)
        x = (x / pow2).clamp(-2 ** (self.bits - 1), 2 ** (self.bits - 1) - 1)
        return ((x.round() - x).detach() + x) * pow2

    def refresh(self):
        """What INT_Quantizer.forward does to its buffers when the rounded exponent changes (quantizers.py:67-71)."""
        with torch.no_grad():
            l = torch.round(torch.log2(self.scale.detach().abs().cpu()))
            dec = l.abs().int().float()
            if float(dec) != float(self.decimal_num):
                self.pow2_scale.copy_(2 ** l)
                self.decimal_num.copy_(dec)
                self.integer_num.copy_(self.bits - 1 - dec)


class _QLinear(nn.Module):
    """INT_Linear state (quant_layers.py:48-85): weight, bias, n_bits buffers, weight/act/out quantisers."""

    def __init__(self, in_features, out_features, bits_w, bits_a, bias=True):
        super().__init__()
        lin = nn.Linear(in_features, out_features, bias=bias)       # same default-init RNG draws as INT_Linear.__init__
        self.weight = lin.weight
        if bias:
            self.bias = lin.bias
        self.register_buffer("n_bits_w", torch.Tensor([bits_w]))
        self.register_buffer("n_bits_a", torch.Tensor([bits_a]))
        self.weight_quantizer = _QScale(bits_w, 2.0 ** (2 - bits_w))
        self.act_quantizer = _QScale(bits_a, 2.0 ** (2 - bits_a))
        self.out_quantizer = _QScale(16, 2.0 ** (2 - 16))
        self.out_quant = False

    def forward(self, x):
        """INT_Linear.forward (quant_layers.py:68-80) through ATen"""
        out = nn.functional.linear(self.act_quantizer.quantise(x), self.weight_quantizer.quantise(self.weight), getattr(self, "bias", None))
        return self.out_quantizer.quantise(out) if (self.out_quant and not self.training) else out


class _QConv2d(nn.Module):
    """INT_Conv2D state (quant_layers.py:10-45): the float layer's weight, a freshly drawn default-init bias (the constructor builds a new
    nn.Conv2d and takes over only the weight), n_bits buffers, weight / activation quantisers — the weight scale starts at
    2 mean|w| / sqrt(Qp) (init_step_size: a 0-dim parameter), no output quantiser."""

    def __init__(self, conv, bits_w, bits_a):
        super().__init__()
        fresh = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, stride=conv.stride, padding=conv.padding,
                          dilation=conv.dilation, groups=conv.groups, bias=conv.bias is not None)       # same RNG draws as INT_Conv2D.__init__
        self.weight = nn.Parameter(conv.weight.detach().clone())
        if conv.bias is not None:
            self.bias = fresh.bias
        self.register_buffer("n_bits_w", torch.Tensor([bits_w]))
        self.register_buffer("n_bits_a", torch.Tensor([bits_a]))
        self.weight_quantizer = _QScale(bits_w, 1.0)
        self.weight_quantizer.scale = nn.Parameter(conv.weight.detach().abs().mean() * 2 / (2 ** (bits_w - 1) - 1) ** 0.5)
        self.act_quantizer = _QScale(bits_a, 2.0 ** (2 - bits_a))
        self.geometry = (conv.stride, conv.padding, conv.dilation, conv.groups)

    def forward(self, x):
        """INT_Conv2D.forward (quant_layers.py:36-45) through ATen"""
        stride, padding, dilation, groups = self.geometry
        return nn.functional.conv2d(self.act_quantizer.quantise(x), self.weight_quantizer.quantise(self.weight), getattr(self, "bias", None),
                                    stride, padding, dilation, groups)


class _QOp(nn.Module):
    def __init__(self, bits):
        super().__init__()
        self.quantizer = _QScale(bits, 2.0 ** (2 - bits))


class _QCell(nn.Module):
    def __init__(self, input_size, hidden_size, bits_w, bits_a):
        super().__init__()
        self.x2h = _QLinear(input_size, 3 * hidden_size, bits_w, bits_a)
        self.h2h = _QLinear(hidden_size, 3 * hidden_size, bits_w, bits_a)
        self.sigmoid, self.tanh, self.add, self.mul = _QOp(bits_a), _QOp(bits_a), _QOp(bits_a), _QOp(bits_a)


class _QRnn(nn.Module):
    def __init__(self, input_size, hidden_size, bits_w, bits_a):
        super().__init__()
        self.rnn_cell_list = nn.ModuleList([_QCell(input_size, hidden_size, bits_w, bits_a)])


class _QuantBase(NativeBackbone):
    """What the quantised backbones share: the optimiser's skip mask, the mode flag, the checkpoint buffers."""

    def _finish(self, hidden_size, bits_w, bits_a, thx=0.0, thh=0.0):
        self.n_bits_w, self.n_bits_a = bits_w, bits_a
        self._finalize(hidden_size, thx, thh, bits_w=bits_w, bits_a=bits_a)
        # AdamW skips parameters whose grad is None: the out_quantizer scales never enter the train-mode graph
        self.frozen_mask = torch.tensor(np.concatenate([np.full(p.numel(), "out_quantizer" in n) for n, p in self.named_parameters()]))
        self._names = [n for n, _ in self.named_parameters()]

    def forward(self, x, h_0=None):
        self.sync_mode()
        return super().forward(x, h_0)

    def sync_mode(self):
        """Before every kernel call on this model (autograd path and the direct-ABI train steps alike): the descriptor's
        ODPD_FLAG_EVAL follows the module's mode — fc_out's 16-bit output quantiser is active in eval only
        (quant_layers.py:77-80) — and the quantisers this call exercises are remembered for the checkpoint buffers."""
        self.desc.flags = (self.desc.flags & ~_lib.FLAG_EVAL) | (0 if self.training else _lib.FLAG_EVAL)
        for name, m in self.named_modules():
            if isinstance(m, _QScale):
                # only fc_out has out_quant set (quant_envs.py:278-287), and its output quantiser runs in eval mode only
                if not name.endswith("out_quantizer") or (name == "fc_out.out_quantizer" and not self.training):
                    m.exercised = True

    def refresh_buffers(self):
        """pow2_scale / decimal_num / integer_num are side effects of INT_Quantizer.forward (quantizers.py:67-71): a quantiser that
        no forward has exercised keeps its construction-time values, as in the reference's checkpoints.  (They are refreshed
        here, from the current scale, rather than at the forward itself — which would cost a device sync per step; the two
        differ only if the rounded exponent flips between the last forward and the save.)"""
        for m in self.modules():
            if isinstance(m, _QScale) and m.exercised:
                m.refresh()

    def state_dict(self, *args, **kwargs):
        self.refresh_buffers()
        return super().state_dict(*args, **kwargs)


_CELL_FEATURES = {"gru": 2, "dgru": 6, "qgru": 4, "qgru_amp1": 4}


class QuantGRUCellModel(_QuantBase):
    """gru / dgru / qgru / qgru_amp1 after the surgery: nn.GRU -> GRU of GRUCells (quant_envs.py:114-130), its Linears and the
    backbone's fc_out (dgru: and fc_hid) -> INT_Linear (:290-306).  Parameter registration order = the reference's
    named_parameters(): rnn, fc_out, fc_hid (dgru.py:22-32)."""

    def __init__(self, backbone_name, hidden_size, bits_w, bits_a):
        super().__init__()
        self.backbone_name = backbone_name
        F = _CELL_FEATURES[backbone_name]
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, F, 2, 1
        self.rnn = _QRnn(F, hidden_size, bits_w, bits_a)
        self.fc_out = _QLinear(hidden_size + (6 if backbone_name == "dgru" else 0), 2, bits_w, bits_a)
        self.fc_out.out_quant = True                                 # quant_envs.py:304
        if backbone_name == "dgru":
            self.fc_hid = _QLinear(hidden_size, hidden_size, bits_w, bits_a)
        self._finish(hidden_size, bits_w, bits_a)


QuantQGRU = QuantGRUCellModel      # the r01 / r02 name


class QuantHeadVDLSTM(_QuantBase):
    """vdlstm after the surgery: fc_lambda_1, fc_lambda_2 and fc_out (named_children order, vdlstm.py:29-43) become INT_Linear, each with its
    own three scales; `set_last_layer_quant` marks fc_out.  nn.LSTM stays float.  Kernels: csrc/lstm_family.hip <VD, QH>."""
    backbone_name = "vdlstm"

    def __init__(self, rnn, bits_w, bits_a):
        super().__init__()
        hidden_size = rnn.hidden_size
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, 4, 2, 1
        self.window_length, self.stride, self.pad_size = 4, 1, 3
        self.rnn = rnn
        self.fc_lambda_1 = _QLinear(hidden_size, 4, bits_w, bits_a)
        self.fc_lambda_2 = _QLinear(hidden_size, 4, bits_w, bits_a)
        self.fc_out = _QLinear(8, 2, bits_w, bits_a)
        self.fc_out.out_quant = True
        self._finish(hidden_size, bits_w, bits_a)


class QuantHeadLSTM(_QuantBase):
    """lstm after the surgery: it holds no nn.GRU and no op modules, so only `fc_out` changes — nn.Linear -> INT_Linear
    (quant_envs.py:40-60, 290-306) with out_quant set (`set_last_layer_quant`, :278-287); nn.LSTM stays float.  Kernels: the
    quantised-head instantiations of csrc/lstm_family.hip (lstm_eval_kernel / lstm_gp_train_kernel / lstm_bwd_kernel <.., QH>)."""
    backbone_name = "lstm"

    def __init__(self, rnn, bits_w, bits_a):
        """`rnn`: the float model's parameter holder (taken over as it is: the surgery deep-copies nn.LSTM, no RNG draws)."""
        super().__init__()
        hidden_size = rnn.hidden_size
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, 2, 2, 1
        self.rnn = rnn
        self.fc_out = _QLinear(hidden_size, 2, bits_w, bits_a)
        self.fc_out.out_quant = True
        self._finish(hidden_size, bits_w, bits_a)


class QuantHeadDeltaJANET(_QuantBase):
    """deltajanet after the surgery: the cell's gates are nn.Parameter tensors of DeltaJANETLayer (deltajanet.py:100-113), so the one module
    the surgery finds is `fc_out` -> INT_Linear with out_quant set.  The sparsity counters and their report stay the float layer's
    (:143-151).  Kernels: the quantised-head instantiations of csrc/deltajanet_wide.hip (one sequence per wave; every hidden size <= 64)."""
    backbone_name = "deltajanet"

    def __init__(self, rnn, bits_w, bits_a, thx=0.0, thh=0.0):
        super().__init__()
        from .backbones.deltagru import _DeltaStats
        hidden_size = rnn.hidden_size
        self.hidden_size, self.input_size, self.output_size, self.num_layers, self.bias = hidden_size, 6, 2, 1, True
        self.rnn = rnn
        self.fc_out = _QLinear(hidden_size, 2, bits_w, bits_a)
        self.fc_out.out_quant = True
        self._dstats = _DeltaStats()
        self.debug = 1
        self._finish(hidden_size, bits_w, bits_a)      # the layer is built with thx = thh = 0 whatever the model was given (:23-27)
        self.thx, self.thh = thx, thh

    def _stats_buffer(self, device):
        return self._dstats.buffer(device) if self.debug else None

    def set_debug(self, value):
        self.debug = value
        self._dstats.reset()

    @property
    def statistics(self):
        return self._dstats.as_dict()

    def get_temporal_sparsity(self):
        st, out = self._dstats.as_dict(), {}
        if self.debug and st["num_dx_numel"] > 0:
            out["SP_T_DX"] = float(st["num_dx_zeros"] / st["num_dx_numel"])
            out["SP_T_DH"] = float(st["num_dh_zeros"] / st["num_dh_numel"])
            out["SP_T_DV"] = float((st["num_dx_zeros"] + st["num_dh_zeros"]) / (st["num_dx_numel"] + st["num_dh_numel"]))
        return out


class QuantHeadNeuralTX(_QuantBase):
    """neuraltx after the surgery: its layer map holds nn.Conv2d and nn.Linear (quant_envs.py:145-148), so the Conv1d FIRs and the Conv1d /
    Hardswish stack stay float and `IQ_match` becomes a bias-free INT_Linear.  No module is named fc_out: set_last_layer_quant (:276-284)
    marks nothing, the 16-bit output quantiser exists (state dict, skipped by the optimiser) but never runs.  Kernels: csrc/tcnn.hip <.., NTX>
    with the descriptor's bits_w > 0."""
    backbone_name = "neuraltx"

    def __init__(self, conv_I, conv_Q, network, bits_w, bits_a):
        super().__init__()
        C = self.hidden_channels = network[0].out_channels
        self.in_channels, self.out_channels, self.kernel_size, self.window_size = 4, 2, 5, 5
        self.conv_I, self.conv_Q, self.network = conv_I, conv_Q, network
        self.IQ_match = _QLinear(2, 2, bits_w, bits_a, bias=False)
        self._finish(C, bits_w, bits_a)


class QuantRVTDCNN(_QuantBase):
    """rvtdcnn after the surgery (feed-forward: every layer is in the surgery's map, quant_envs.py:145-148): Conv2d -> INT_Conv2D, fc_hid and
    fc_out -> INT_Linear (fc_out with out_quant: set_last_layer_quant), the functional tanh calls stay float.  Kernels: csrc/rvtdcnn_q.hip."""
    backbone_name = "rvtdcnn"

    def __init__(self, conv, fc_hid_size, bits_w, bits_a):
        super().__init__()
        self.window_size, self.out_channels, self.fc_hid_size = 4, 3, fc_hid_size
        self.stride, self.feature_size_new, self.fc_in_features = 1, 3, 36
        self.Conv2d = _QConv2d(conv, bits_w, bits_a)             # (the surgery walks the layer map type by type: Conv2d first, then the Linears)
        self.fc_hid = _QLinear(36, fc_hid_size, bits_w, bits_a)
        self.fc_out = _QLinear(fc_hid_size, 2, bits_w, bits_a)
        self.fc_out.out_quant = True
        self._finish(fc_hid_size, bits_w, bits_a)


class QuantPGJANET(_QuantBase):
    """pgjanet after the surgery: its six nn.Linear — the gates W_a, W_p1, W_p2, W_f, W_g and the read-out W_o (pgjanet.py:11-21) — become
    INT_Linear in named_children order, each with three scales; the functional tanh / sigmoid calls stay float; no module is named fc_out, so
    no output quantiser runs.  Kernels: csrc/pgjanet_q.hip (hidden <= 32)."""
    backbone_name = "pgjanet"

    def __init__(self, hidden_size, bits_w, bits_a):
        super().__init__()
        H = self.hidden_size = hidden_size
        self.output_size, self.bias = 2, True
        self.W_a, self.W_p1, self.W_p2 = (_QLinear(H + 1, H, bits_w, bits_a) for _ in range(3))
        self.W_f, self.W_g = _QLinear(2 * H, H, bits_w, bits_a), _QLinear(2 * H, H, bits_w, bits_a)
        self.W_o = _QLinear(H, 2, bits_w, bits_a)
        self._finish(H, bits_w, bits_a)


class _QDeltaLayer(nn.Module):
    """DeltaGRULayer of deltagru_tcnskip.py:133-162 after the surgery: bias-free INT_Linear x2h / h2h, Quant_add / mult / sigmoid /
    tanh in the layer's own registration order."""

    def __init__(self, hidden_size, bits_w, bits_a):
        super().__init__()
        self.x2h = _QLinear(6, 3 * hidden_size, bits_w, bits_a, bias=False)
        self.h2h = _QLinear(hidden_size, 3 * hidden_size, bits_w, bits_a, bias=False)
        self.add, self.mul, self.sigmoid, self.tanh = _QOp(bits_a), _QOp(bits_a), _QOp(bits_a), _QOp(bits_a)


class _Weight(nn.Module):
    """A convolution's weight without the convolution (and without its default-init RNG draws: the reference deep-copies these)."""

    def __init__(self, *shape):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(*shape))


class QuantTResDeltaGRU(_QuantBase):
    """deltagru_tcnskip after the surgery (the OpenDPDv2 QAT stage, bash_scripts/OpenDPDv2.sh:84-117): quantised delta cell and
    fc_out, float TCN skip (Conv1d / Hardswish are not swapped).  Same thresholds / sparsity interface as the float backbone."""
    backbone_name = "deltagru_tcnskip"
    fused_stats = True      # the one-launch train step counts the sparsity statistics through its `workspace` argument (train_funcs.train_workspace)

    def __init__(self, hidden_size, bits_w, bits_a, thx=0.0, thh=0.0):
        super().__init__()
        from .backbones.deltagru import _DeltaStats
        self.hidden_size, self.input_size, self.output_size, self.num_layers, self.bias = hidden_size, 6, 2, 1, True
        self.rnn = _QDeltaLayer(hidden_size, bits_w, bits_a)
        self.fc_out = _QLinear(hidden_size, 2, bits_w, bits_a, bias=False)
        self.fc_out.out_quant = True
        self.tcn = nn.Sequential(_Weight(3, 2, 3), nn.Identity(), _Weight(2, 3, 1), nn.Identity())
        self.thx, self.thh = thx, thh
        self._dstats = _DeltaStats()
        self.debug = 1
        self._finish(hidden_size, bits_w, bits_a, thx, thh)

    def _stats_buffer(self, device):
        return self._dstats.buffer(device) if self.debug else None

    def set_debug(self, value):
        self.debug = value
        self._dstats.reset()

    @property
    def statistics(self):
        return self._dstats.as_dict()

    def get_temporal_sparsity(self):
        """deltagru_tcnskip.py:105-129 on the quantised module: the 'weight' / 'bias' name tests see the quantiser scales too."""
        st = self._dstats.as_dict()
        out = {}
        if self.debug and st["num_dx_numel"] > 0:
            rnn_w = sum(p.numel() for n, p in self.rnn.named_parameters() if "weight" in n)
            rnn_b = sum(p.numel() for n, p in self.rnn.named_parameters() if "bias" in n)
            fc = sum(p.numel() for p in self.fc_out.parameters()) + sum(p.numel() for p in self.tcn.parameters())
            tz, tn = st["num_dx_zeros"] + st["num_dh_zeros"], st["num_dx_numel"] + st["num_dh_numel"]
            out["SP_T_DX"] = float(st["num_dx_zeros"] / st["num_dx_numel"])
            out["SP_T_DH"] = float(st["num_dh_zeros"] / st["num_dh_numel"])
            out["SP_T_DV"] = float(tz / tn)
            out["HW_PARAM"] = float(fc + rnn_w * (1 - float(tz / tn)) + rnn_b)
        return out


MAX_HIDDEN = 32          # csrc/qat_s16.hip: two 16-unit tiles
_UNTOUCHED = ("gmp", "tcnn")       # no nn.GRU, no nn.Linear, no op modules: the surgery returns an identical deep copy
_PARTIAL = ("apnrru", "bojanet", "dvrjanet", "mcldnn")
_HEAD_ONLY = ("lstm", "vdlstm", "deltajanet", "neuraltx", "rvtdcnn", "pgjanet")    # only nn.Linear / nn.Conv2d layers to swap, and kernels for the result exist
_HEAD_MAX_HIDDEN = {"deltajanet": 64, "neuraltx": 64, "lstm": 64}      # csrc/deltajanet_wide.hip, tcnn.hip, lstm_wide.hip (33 .. 64) carry the quantised head
_HEAD_LAYERS = {"pgjanet": ("W_a", "W_p1", "W_p2", "W_f", "W_g", "W_o"), "rvtdcnn": ("Conv2d", "fc_hid", "fc_out"), "lstm": ("fc_out",), "vdlstm": ("fc_lambda_1", "fc_lambda_2", "fc_out"), "deltajanet": ("fc_out",), "neuraltx": ("IQ_match",)}
_FLOAT_CORE = {"neuraltx": ("conv_I", "conv_Q", "network"), "rvtdcnn": (), "pgjanet": ()}   # (the others: "rnn")


def _warn_float(exc, model):
    print(f"[WARN] Quantization setup failed: {exc}. Using float model instead.")
    return model


def _wrap(model, bb, dev):
    q = CoreModel.__new__(CoreModel)
    nn.Module.__init__(q)
    for k in ("output_size", "input_size", "hidden_size", "num_layers", "backbone_type", "thx", "thh", "window_size",
              "num_dvr_units", "batch_first", "bidirectional", "bias"):
        setattr(q, k, getattr(model, k))
    q.backbone = bb
    return q.to(dev)


def _quantise_heads(model, bits_w, bits_a, pre, dev):
    """lstm / vdlstm / deltajanet / neuraltx: create_pygru_model finds no nn.GRU (no RNG draws), load_model strict-loads a float checkpoint of the model's own keys,
    create_quantized_model swaps the nn.Linear heads in named_children order — each INT_Linear keeps the weight and draws a fresh
    default-init bias (quant_layers.py:48-56)."""
    import copy
    fb = model.backbone
    if pre:
        try:
            pre_sd = torch.load(pre, map_location="cpu")
            want = {"backbone." + k: tuple(v.shape) for k, v in fb.state_dict().items()}
            if not isinstance(pre_sd, dict) or set(pre_sd) != set(want) or any(tuple(pre_sd[k].shape) != s for k, s in want.items()):
                raise RuntimeError("Error(s) in loading state_dict for CoreModel")
        except Exception as exc:
            return _warn_float(exc, model)
    heads = _HEAD_LAYERS[model.backbone_type]
    with torch.no_grad():
        core = {c: copy.deepcopy(getattr(fb, c)).cpu() for c in _FLOAT_CORE.get(model.backbone_type, ("rnn",))}
        fc_w = {h: getattr(fb, h).weight.detach().cpu() for h in heads}
        if pre:
            for c, mod in core.items():
                for k, p in mod.named_parameters():
                    p.copy_(pre_sd[f"backbone.{c}.{k}"])
            fc_w = {h: pre_sd[f"backbone.{h}.weight"] for h in heads}
        rnn = core.get("rnn")
        if model.backbone_type == "pgjanet":
            bb = QuantPGJANET(fb.hidden_size, bits_w, bits_a)
        elif model.backbone_type == "rvtdcnn":
            conv = copy.deepcopy(fb.Conv2d).cpu()
            conv.weight.copy_(fc_w["Conv2d"])
            bb = QuantRVTDCNN(conv, fb.fc_hid_size, bits_w, bits_a)
        elif model.backbone_type == "neuraltx":
            bb = QuantHeadNeuralTX(core["conv_I"], core["conv_Q"], core["network"], bits_w, bits_a)
        elif model.backbone_type == "deltajanet":
            bb = QuantHeadDeltaJANET(rnn, bits_w, bits_a, model.thx, model.thh)
        else:
            bb = (QuantHeadVDLSTM if model.backbone_type == "vdlstm" else QuantHeadLSTM)(rnn, bits_w, bits_a)
        for h in heads:
            getattr(bb, h).weight.copy_(fc_w[h])
    return _wrap(model, bb, dev)


def _aten_swap(mod, layer_type, bits_w, bits_a):
    """recur_rpls_layers (quant_envs.py:40-60) for one layer type: named_children order, depth first; a replaced layer draws its fresh default
    initialisation exactly where INT_Linear / INT_Conv2D's constructors draw theirs and keeps the float layer's weight"""
    for name, child in list(mod.named_children()):
        if type(child) is layer_type:
            if layer_type is nn.Linear:
                q = _QLinear(child.in_features, child.out_features, bits_w, bits_a, bias=child.bias is not None)
                with torch.no_grad():
                    q.weight.copy_(child.weight)
            else:
                q = _QConv2d(child, bits_w, bits_a)
            setattr(mod, name, q)
        else:
            _aten_swap(child, layer_type, bits_w, bits_a)


def _quantise_aten(model, bits_w, bits_a, pre, dev):
    """apnrru / bojanet / dvrjanet / mcldnn (quant_envs.py:285-306 runs on them like on every registry model): their gates, FIR banks and
    read-outs are nn.Linear (mcldnn: two nn.Conv2d and two nn.Linear next to a float Conv1d and nn.LSTM) — all of them become INT_Linear /
    INT_Conv2D, the functional sigmoid / tanh calls stay float.  There are no HIP kernels for a quantised mat-vec INSIDE these cells: the
    quantised model is the ATen restatement of the backbone (backbones/extras.py) with the surgery applied to it — `native` False, said
    aloud once per configuration, torch optimiser — pinned to fixtures produced by the reference (tests/test_quant_partial_cpu.py)."""
    import warnings
    from .backbones import extras as X
    bt, H = model.backbone_type, model.hidden_size
    fb = model.backbone
    sd = {k: v.detach().cpu() for k, v in fb.state_dict().items()}
    if pre:
        try:      # load_model strict-loads the float checkpoint before the layers are swapped (quant_envs.py:173-182)
            pre_sd = torch.load(pre, map_location="cpu")
            want = {"backbone." + k: tuple(v.shape) for k, v in sd.items()}
            if not isinstance(pre_sd, dict) or set(pre_sd) != set(want) or any(tuple(pre_sd[k].shape) != s_ for k, s_ in want.items()):
                raise RuntimeError("Error(s) in loading state_dict for CoreModel")
            sd = {k: pre_sd["backbone." + k] for k in sd}
        except Exception as exc:
            return _warn_float(exc, model)
    rng = torch.get_rng_state()      # building the holder draws initialisations the reference (a deepcopy) does not: keep the stream where it was
    ext = {"apnrru": lambda: X.APNRRU(hidden_size=H, bias=True), "bojanet": lambda: X.BOJANET(hidden_size=H, output_size=2, bias=True),
           "dvrjanet": lambda: X.DVRJANET(hidden_size=H, output_size=2, num_dvr_units=model.num_dvr_units, bias=True),
           "mcldnn": lambda: X.MCLDNN(hidden_size=H)}[bt]()
    torch.set_rng_state(rng)
    ext.load_state_dict(sd)
    for layer_type in (nn.Conv2d, nn.Linear):       # the order of fq_layers_hash (quant_envs.py:145-148)
        _aten_swap(ext, layer_type, bits_w, bits_a)
    for name, mod in ext.named_modules():            # set_last_layer_quant: every module NAMED fc_out (quant_envs.py:276-284)
        if name.split(".")[-1] == "fc_out" and isinstance(mod, _QLinear):
            mod.out_quant = True
    warnings.warn(f"opendpd_amd: --quant on '{bt}' runs the ATen restatement of the quantised model (backbones/extras.py + INT_Linear / INT_Conv2D "
                  "through torch ops): there are no HIP kernels for quantised mat-vecs inside this cell", stacklevel=3)
    return _wrap(model, ext, dev)


def get_quant_model(proj, model):
    """Reference semantics (quant/__init__.py:20-37): identity unless `proj.quant`; otherwise the quantised model.
    `proj` needs n_bits_w, n_bits_a and optionally pretrained_model.

    The surgery is generic (quant_envs.py:114-130, 290-306).  HIP-backed here: gru, dgru, qgru, qgru_amp1 (GRU of GRUCells, INT_Linear
    heads) and deltagru_tcnskip (its layer's Linears and op modules), one layer, hidden <= 32; gmp and tcnn contain nothing the surgery
    swaps (the reference hands back an identical copy: the model itself is returned).  deltagru's layer IS an nn.GRU subclass, the
    reference swaps it for a plain GRU and then fails in forward (TypeError, deltagru.py:74-77): refused here at construction.  In lstm,
    vdlstm, deltajanet and neuraltx the surgery finds only nn.Linear HEADS (float core, INT_Linear heads: `_quantise_heads`; deltajanet
    and neuraltx up to 64 units / channels); the backbones whose gates or convolutions are themselves nn.Linear / nn.Conv2d modules INSIDE a recurrent
    cell (`_PARTIAL`: apnrru, bojanet, dvrjanet, mcldnn) have no quantised kernels: their quantised model is the ATen restatement of the backbone
    with the surgery applied (`_quantise_aten`: `native` False, announced by a warning) — it trains, as it does in the reference.

    `pretrained_model` follows Base_GRUQuantEnv.load_model (quant_envs.py:173-182): the checkpoint is strict-loaded into the FLOAT
    holder before quantisation — for the GRU-cell models its keys are `backbone.rnn.rnn_cell_list.0.{x2h,h2h}.{weight,bias}`,
    `backbone.fc_out.*` (+ `backbone.fc_hid.*`), for deltagru_tcnskip the float model's own keys (so a float `train_dpd` checkpoint
    loads: the OpenDPDv2 flow).  INT_Linear then keeps the weights and draws fresh biases (quant_layers.py:48-56), the scales start at
    their defaults.  Any failure on the way (unreadable file, other key set, other shapes) raises inside the reference's try block:
    it warns and returns the float model it was given (quant/__init__.py:35-37), after the RNG draws made up to that point — the
    same happens here (tests/golden/quant_pretrained_qgru_h10.npz)."""
    if not getattr(proj, "quant", False):
        return model
    bt = getattr(model, "backbone_type", None)
    if not isinstance(model, CoreModel) or bt is None:
        raise NotImplementedError("quantisation-aware training takes a CoreModel")
    if bt in _UNTOUCHED:
        return model
    if bt == "deltagru":
        raise RuntimeError("--quant on 'deltagru': its layer subclasses nn.GRU, the reference's surgery replaces it by a plain GRU and the "
                           "model then fails in forward (TypeError); use 'deltagru_tcnskip'")
    bits_w, bits_a = int(getattr(proj, "n_bits_w", 8)), int(getattr(proj, "n_bits_a", 8))
    if bt in _PARTIAL:
        return _quantise_aten(model, bits_w, bits_a, getattr(proj, "pretrained_model", ""), next(model.parameters()).device)
    H = model.hidden_size
    max_h = _HEAD_MAX_HIDDEN.get(bt, MAX_HIDDEN)
    # pgjanet, rvtdcnn and neuraltx ignore num_layers (models.py:26-141 never hands it to them; wide.outside_envelope treats them the same way):
    # `--quant --DPD_num_layers 2` on them runs in the reference and must run here; the check is for the recurrent cores
    layers_matter = bt not in ("pgjanet", "rvtdcnn", "neuraltx")
    if H > max_h or (layers_matter and model.num_layers != 1):
        raise NotImplementedError(f"the QAT kernels cover one layer and hidden_size <= {max_h} (csrc/qat_s16.hip; csrc/deltajanet_wide.hip)")
    dev = next(model.parameters()).device
    pre = getattr(proj, "pretrained_model", "")
    if bt in _HEAD_ONLY:
        return _quantise_heads(model, bits_w, bits_a, pre, dev)
    tres = bt == "deltagru_tcnskip"
    holder = None
    if not tres:
        # --- RNG consumption order of Base_GRUQuantEnv (quant_envs.py:156-171, 198-246, 290-306) ----------------------
        # 1. recur_rpls_gru: PYGRU -> GRUCell(F,H): two nn.Linear default inits, then GRUCell.reset_parameters (uniform over
        #    x2h.weight, x2h.bias, h2h.weight, h2h.bias — quant/modules/gru.py:24-29)
        F = _CELL_FEATURES[bt]
        std = 1.0 / math.sqrt(H)
        holder = nn.Module()
        holder.x2h, holder.h2h = nn.Linear(F, 3 * H), nn.Linear(H, 3 * H)
        for w in holder.parameters():
            nn.init.uniform_(w, -std, std)
        # 2. _reset_pygru: biases 0, gate blocks orthogonal, x2h.weight gate blocks xavier (quant_envs.py:205-227)
        init_gatewise(holder, H, xavier_suffix="x2h.weight")
    pre_sd = None
    if pre:
        # load_model runs after create_pygru_model, inside get_quant_model's try block
        try:
            pre_sd = torch.load(pre, map_location="cpu")
        except Exception as exc:
            return _warn_float(exc, model)
        if tres:
            want = {"backbone.rnn.x2h.weight": (3 * H, 6), "backbone.rnn.h2h.weight": (3 * H, H), "backbone.fc_out.weight": (2, H),
                    "backbone.tcn.0.weight": (3, 2, 3), "backbone.tcn.2.weight": (2, 3, 1)}
        else:
            cellp = "backbone.rnn.rnn_cell_list.0."
            OW = H + 6 if bt == "dgru" else H
            want = {cellp + "x2h.weight": (3 * H, F), cellp + "x2h.bias": (3 * H,), cellp + "h2h.weight": (3 * H, H),
                    cellp + "h2h.bias": (3 * H,), "backbone.fc_out.weight": (2, OW), "backbone.fc_out.bias": (2,)}
            if bt == "dgru":
                want.update({"backbone.fc_hid.weight": (H, H), "backbone.fc_hid.bias": (H,)})
        bad = not isinstance(pre_sd, dict) or set(pre_sd) != set(want) or any(tuple(pre_sd[k].shape) != s for k, s in want.items())
        if bad:
            keys = set(pre_sd) if isinstance(pre_sd, dict) else set()
            missing, extra = sorted(set(want) - keys), sorted(keys - set(want))
            return _warn_float(f"Error(s) in loading state_dict for CoreModel: missing {missing[:4]}, unexpected {extra[:4]}"
                               f"{' ...' if len(extra) > 4 else ''}", model)
    # 3. create_quantized_model: INT_Linear(m) for every nn.Linear in named_children order (x2h, h2h, fc_out, fc_hid): each draws a
    #    fresh default nn.Linear init and keeps only m.weight — a bias stays the freshly drawn one (quant_layers.py:48-56)
    fb = model.backbone
    with torch.no_grad():
        if tres:
            bb = QuantTResDeltaGRU(H, bits_w, bits_a, thx=model.thx, thh=model.thh)
            src = pre_sd if pre_sd is not None else {"backbone." + k: v.detach().cpu() for k, v in fb.state_dict().items()}
            bb.rnn.x2h.weight.copy_(src["backbone.rnn.x2h.weight"])
            bb.rnn.h2h.weight.copy_(src["backbone.rnn.h2h.weight"])
            bb.fc_out.weight.copy_(src["backbone.fc_out.weight"])
            bb.tcn[0].weight.copy_(src["backbone.tcn.0.weight"])
            bb.tcn[2].weight.copy_(src["backbone.tcn.2.weight"])
        else:
            bb = QuantGRUCellModel(bt, H, bits_w, bits_a)
            cell = bb.rnn.rnn_cell_list[0]
            cell.x2h.weight.copy_(holder.x2h.weight)
            cell.h2h.weight.copy_(holder.h2h.weight)
            bb.fc_out.weight.copy_(fb.fc_out.weight.detach().cpu())
            if bt == "dgru":
                bb.fc_hid.weight.copy_(fb.fc_hid.weight.detach().cpu())
            if pre_sd is not None:          # load_model ran between steps 2 and 3: the weights INT_Linear keeps are the checkpoint's
                cell.x2h.weight.copy_(pre_sd[cellp + "x2h.weight"])
                cell.h2h.weight.copy_(pre_sd[cellp + "h2h.weight"])
                bb.fc_out.weight.copy_(pre_sd["backbone.fc_out.weight"])
                if bt == "dgru":
                    bb.fc_hid.weight.copy_(pre_sd["backbone.fc_hid.weight"])
    return _wrap(model, bb, dev)
