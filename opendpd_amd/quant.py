"""Quantisation-aware QGRU: drop-in for the reference's `quant.get_quant_model` (quant/__init__.py:20-37) applied to a
`CoreModel('qgru' | 'qgru_amp1')`.

The reference performs model surgery (quant/quant_envs.py:138-306): nn.GRU -> Python GRU of GRUCells (re-initialised),
nn.Linear -> INT_Linear, Sigmoid/Tanh/Add/Mul -> Quant_* with one power-of-two scale parameter each.  Here the result of
that surgery is a single HIP-backed backbone, `QuantQGRU`, with the same parameter / buffer names
(`backbone.rnn.rnn_cell_list.0.{x2h,h2h}.{weight,bias,weight_quantizer.scale,...}`, `...{sigmoid,tanh,add,mul}.quantizer.scale`,
`backbone.fc_out.*`) so state dicts are interchangeable, and the same construction-time RNG consumption.
Kernels: csrc/qgru_family.hip (integer-grid arithmetic bit-exact with the reference for 8-bit grids)."""
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

    def __init__(self, in_features, out_features, bits_w, bits_a):
        super().__init__()
        lin = nn.Linear(in_features, out_features, bias=True)       # same default-init RNG draws as INT_Linear.__init__
        self.weight, self.bias = lin.weight, lin.bias
        self.register_buffer("n_bits_w", torch.Tensor([bits_w]))
        self.register_buffer("n_bits_a", torch.Tensor([bits_a]))
        self.weight_quantizer = _QScale(bits_w, 2.0 ** (2 - bits_w))
        self.act_quantizer = _QScale(bits_a, 2.0 ** (2 - bits_a))
        self.out_quantizer = _QScale(16, 2.0 ** (2 - 16))
        self.out_quant = False


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


class QuantQGRU(NativeBackbone):
    """Quantised qgru / qgru_amp1 backbone."""

    def __init__(self, backbone_name, hidden_size, bits_w, bits_a):
        super().__init__()
        self.backbone_name = backbone_name
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, 4, 2, 1
        self.n_bits_w, self.n_bits_a = bits_w, bits_a
        self.rnn = _QRnn(4, hidden_size, bits_w, bits_a)
        self.fc_out = _QLinear(hidden_size, 2, bits_w, bits_a)
        self.fc_out.out_quant = True                                 # quant_envs.py:304
        self._finalize(hidden_size, bits_w=bits_w, bits_a=bits_a)
        names = [n for n, _ in self.named_parameters()]
        # AdamW skips parameters whose grad is None: the out_quantizer scales never enter the train-mode graph
        self.frozen_mask = torch.tensor(np.concatenate([np.full(p.numel(), "out_quantizer" in n)
                                                        for n, p in self.named_parameters()]))
        self._names = names

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
                # x2h / h2h never use their out_quantizer (out_quant is False there), fc_out's runs in eval mode only
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


def get_quant_model(proj, model):
    """Reference semantics (quant/__init__.py:20-37): identity unless `proj.quant`; otherwise the quantised model.
    `proj` needs n_bits_w, n_bits_a and optionally pretrained_model.

    `pretrained_model` follows Base_GRUQuantEnv.load_model (quant_envs.py:173-182): the checkpoint is strict-loaded into the FLOAT
    holder (keys `backbone.rnn.rnn_cell_list.0.{x2h,h2h}.{weight,bias}`, `backbone.fc_out.{weight,bias}`) before quantisation;
    INT_Linear then keeps the weights and draws fresh biases (quant_layers.py:48-56), the scales start at their defaults.  Any other
    key set — the nn.GRU names of a float `train_dpd` checkpoint, or a quantised checkpoint with its scales and buffers — makes that
    strict load raise in the reference, whose get_quant_model then warns and returns the float model it was given
    (quant/__init__.py:35-37); the same happens here (pinned by tests/golden/quant_pretrained_qgru_h10.npz)."""
    if not getattr(proj, "quant", False):
        return model
    if not isinstance(model, CoreModel) or model.backbone_type not in ("qgru", "qgru_amp1"):
        raise NotImplementedError("quantisation-aware training is implemented for the qgru / qgru_amp1 backbones")
    bits_w, bits_a = int(getattr(proj, "n_bits_w", 8)), int(getattr(proj, "n_bits_a", 8))
    H = model.hidden_size
    if H > 16 or model.num_layers != 1:
        raise NotImplementedError("the QAT kernels cover one layer and hidden_size <= 16 (csrc/qgru_family.hip)")
    dev = next(model.parameters()).device
    pre = getattr(proj, "pretrained_model", "")
    pre_sd = None
    if pre:
        pre_sd = torch.load(pre, map_location="cpu")
        cellp = "backbone.rnn.rnn_cell_list.0."
        want = {cellp + "x2h.weight": (3 * H, 4), cellp + "x2h.bias": (3 * H,), cellp + "h2h.weight": (3 * H, H),
                cellp + "h2h.bias": (3 * H,), "backbone.fc_out.weight": (2, H), "backbone.fc_out.bias": (2,)}
        bad = set(pre_sd) != set(want) or any(tuple(pre_sd[k].shape) != s for k, s in want.items())
        if bad:
            missing, extra = sorted(set(want) - set(pre_sd)), sorted(set(pre_sd) - set(want))
            print(f"[WARN] Quantization setup failed: Error(s) in loading state_dict for CoreModel: missing {missing[:4]}, "
                  f"unexpected {extra[:4]}{' ...' if len(extra) > 4 else ''}. Using float model instead.")
            return model
    # --- RNG consumption order of Base_GRUQuantEnv (quant_envs.py:156-171, 198-246, 290-306) ----------------------
    # 1. recur_rpls_gru: PYGRU -> GRUCell(4,H): two nn.Linear default inits, then GRUCell.reset_parameters (uniform over
    #    x2h.weight, x2h.bias, h2h.weight, h2h.bias — quant/modules/gru.py:24-29)
    std = 1.0 / math.sqrt(H)
    holder = nn.Module()
    holder.x2h, holder.h2h = nn.Linear(4, 3 * H), nn.Linear(H, 3 * H)
    for w in holder.parameters():
        nn.init.uniform_(w, -std, std)
    # 2. _reset_pygru: biases 0, gate blocks orthogonal, x2h.weight gate blocks xavier (quant_envs.py:205-227)
    init_gatewise(holder, H, xavier_suffix="x2h.weight")
    # 3. create_quantized_model: INT_Linear(m) for x2h, h2h, then fc_out: each draws a fresh default nn.Linear init and keeps
    #    only m.weight — the bias stays the freshly drawn one (quant_layers.py:48-56)
    bb = QuantQGRU(model.backbone_type, H, bits_w, bits_a)
    cell = bb.rnn.rnn_cell_list[0]
    with torch.no_grad():
        cell.x2h.weight.copy_(holder.x2h.weight)
        cell.h2h.weight.copy_(holder.h2h.weight)
        bb.fc_out.weight.copy_(model.backbone.fc_out.weight.detach().cpu())
        if pre_sd is not None:          # load_model ran between steps 2 and 3: the weights INT_Linear keeps are the checkpoint's
            cell.x2h.weight.copy_(pre_sd[cellp + "x2h.weight"])
            cell.h2h.weight.copy_(pre_sd[cellp + "h2h.weight"])
            bb.fc_out.weight.copy_(pre_sd["backbone.fc_out.weight"])
    q = CoreModel.__new__(CoreModel)
    nn.Module.__init__(q)
    for k in ("output_size", "input_size", "hidden_size", "num_layers", "backbone_type", "thx", "thh", "window_size",
              "num_dvr_units", "batch_first", "bidirectional", "bias"):
        setattr(q, k, getattr(model, k))
    q.backbone = bb
    return q.to(dev)
