"""HIP-backed delta-network GRU backbones: deltagru, deltagru_tcnskip (TRes-DeltaGRU).

Reference behaviour reproduced:
  DeltaGRU            backbones/deltagru.py:10-276 — rnn = DeltaGRULayer(nn.GRU) (keys rnn.weight_ih_l0 ...), fc_out (2,H)+bias;
                      the layer's own reset_parameters (orthogonal weights, zero biases, :141-146) runs inside
                      nn.GRU.__init__, then DeltaGRU.reset_parameters (:42-57).
  DeltaGRU (tcnskip)  backbones/deltagru_tcnskip.py:11-304 — rnn.x2h / rnn.h2h bias-free Linears, fc_out bias-free,
                      tcn = Sequential(Conv1d(2,3,k3,dil16,pad16), Hardswish, Conv1d(3,2,k1), Hardswish).
Both expose thx / thh, set_debug(v) and get_temporal_sparsity() (used by modules/paths.py:49-59).
Kernels: opendpd_amd/csrc/delta_family.hip (sparsity counters are accumulated on the device).
"""
import torch
import torch.nn as nn

from .gru import _check_single_layer
from .native import NativeBackbone, init_gatewise, init_linear


class _DeltaStats:
    """Device-side counters [dx_zeros, dx_numel, dh_zeros, dh_numel] presented like the reference's
    `rnn.statistics` dict."""

    def __init__(self):
        self.buf = None

    def buffer(self, device):
        if self.buf is None or self.buf.device != device:
            self.buf = torch.zeros(4, dtype=torch.float64, device=device)
        return self.buf

    def reset(self):
        if self.buf is not None:
            self.buf.zero_()

    def as_dict(self):
        v = self.buf.cpu().tolist() if self.buf is not None else [0.0, 0.0, 0.0, 0.0]
        return {"num_dx_zeros": v[0], "num_dx_numel": v[1], "num_dh_zeros": v[2], "num_dh_numel": v[3]}


class _GruLayerParams(nn.Module):
    """Parameter holder equal to DeltaGRULayer(nn.GRU) after its constructor (deltagru.py:104-146)."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        G = 3 * hidden_size
        self.weight_ih_l0 = nn.Parameter(torch.empty(G, input_size))
        self.weight_hh_l0 = nn.Parameter(torch.empty(G, hidden_size))
        self.bias_ih_l0 = nn.Parameter(torch.empty(G))
        self.bias_hh_l0 = nn.Parameter(torch.empty(G))
        for name, p in self.named_parameters():
            if "weight" in name:
                nn.init.orthogonal_(p)
            elif "bias" in name:
                nn.init.constant_(p, 0)


class _DeltaBase(NativeBackbone):
    dx_needs_flag = True

    def _setup_delta(self, hidden_size, thx, thh):
        self.thx, self.thh = thx, thh
        self._dstats = _DeltaStats()
        self.debug = 1
        self._finalize(hidden_size, thx, thh)

    def _stats_buffer(self, device):
        return self._dstats.buffer(device) if self.debug else None

    def set_debug(self, value):
        self.debug = value
        self._dstats.reset()

    @property
    def statistics(self):
        return self._dstats.as_dict()

    def _sparsity(self, fc_numel):
        st = self._dstats.as_dict()
        out = {}
        if self.debug and st["num_dx_numel"] > 0:
            rnn_w = sum(p.numel() for n, p in self.rnn.named_parameters() if "weight" in n)
            rnn_b = sum(p.numel() for n, p in self.rnn.named_parameters() if "bias" in n)
            tz, tn = st["num_dx_zeros"] + st["num_dh_zeros"], st["num_dx_numel"] + st["num_dh_numel"]
            out["SP_T_DX"] = float(st["num_dx_zeros"] / st["num_dx_numel"])
            out["SP_T_DH"] = float(st["num_dh_zeros"] / st["num_dh_numel"])
            out["SP_T_DV"] = float(tz / tn)
            out["HW_PARAM"] = float(fc_numel + rnn_w * (1 - float(tz / tn)) + rnn_b)
        return out


class DeltaGRU(_DeltaBase):
    backbone_name = "deltagru"

    def __init__(self, input_size, hidden_size, output_size, num_layers, thx=0, thh=0, bias=True):
        super().__init__()
        _check_single_layer(num_layers, False)
        self.hidden_size, self.input_size, self.output_size, self.num_layers, self.bias = hidden_size, 6, output_size, 1, bias
        self.rnn = _GruLayerParams(6, hidden_size)
        self.fc_out = nn.Linear(hidden_size, output_size, bias=True)
        self._setup_delta(hidden_size, thx, thh)

    def reset_parameters(self):
        init_gatewise(self.rnn, self.hidden_size)
        init_linear(self.fc_out, "xavier")

    def get_temporal_sparsity(self):
        return self._sparsity(sum(p.numel() for p in self.fc_out.parameters()))


class _TresLayerParams(nn.Module):
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.x2h = nn.Linear(input_size, 3 * hidden_size, bias=False)
        self.h2h = nn.Linear(hidden_size, 3 * hidden_size, bias=False)


class TResDeltaGRU(_DeltaBase):
    """Registry name 'deltagru_tcnskip'."""
    backbone_name = "deltagru_tcnskip"

    def __init__(self, input_size, hidden_size, output_size, num_layers, thx=0, thh=0, bias=True):
        super().__init__()
        _check_single_layer(num_layers, False)
        self.hidden_size, self.input_size, self.output_size, self.num_layers, self.bias = hidden_size, 6, output_size, 1, bias
        self.rnn = _TresLayerParams(6, hidden_size)
        self.fc_out = nn.Linear(hidden_size, output_size, bias=False)
        self.tcn = nn.Sequential(nn.Conv1d(2, 3, kernel_size=3, padding=16, stride=1, dilation=16, bias=False), nn.Hardswish(),
                                 nn.Conv1d(3, 2, kernel_size=1, padding=0, stride=1, dilation=1, bias=False), nn.Hardswish())
        self._setup_delta(hidden_size, thx, thh)

    def reset_parameters(self):
        for name, p in self.tcn.named_parameters():
            if "weight" in name:
                nn.init.xavier_uniform_(p)
        init_gatewise(self.rnn, self.hidden_size, xavier_suffix="x2h.weight")
        init_linear(self.fc_out, "xavier")

    def get_temporal_sparsity(self):
        return self._sparsity(sum(p.numel() for p in self.fc_out.parameters()) + sum(p.numel() for p in self.tcn.parameters()))


class _JanetLayerParams(nn.Module):
    """Parameter holder equal to DeltaJANETLayer after its constructor (deltajanet.py:96-141): two gates [f; g], every weight
    orthogonal as a whole (2H x k) matrix, zero biases."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        G = 2 * hidden_size
        self.weight_ih_l0 = nn.Parameter(torch.empty(G, input_size))
        self.weight_hh_l0 = nn.Parameter(torch.empty(G, hidden_size))
        self.bias_ih_l0 = nn.Parameter(torch.empty(G))
        self.bias_hh_l0 = nn.Parameter(torch.empty(G))
        for name, p in self.named_parameters():
            if "weight" in name:
                nn.init.orthogonal_(p)
            elif "bias" in name:
                nn.init.constant_(p, 0)


class DeltaJANET(_DeltaBase):
    """Registry name 'deltajanet' (backbones/deltajanet.py:11-274): features as deltagru, a two-gate delta cell (f and g both sigmoids,
    h = (1-f) g + f h), fc_out with bias.  The reference builds its layer with thx = thh = 0 whatever it is given (:23-27): the
    thresholds are kept as attributes (the train_dpd log reads them) but the kernels run with 0, so nothing is ever masked and the
    counters record exact repeats only.  Kernels: the JAN instantiation of csrc/delta_s16.hip (hidden <= 32, every batch size)."""
    backbone_name = "deltajanet"

    def __init__(self, input_size, hidden_size, output_size, num_layers, thx=0, thh=0, bias=True):
        super().__init__()
        _check_single_layer(num_layers, False)
        self.hidden_size, self.input_size, self.output_size, self.num_layers, self.bias = hidden_size, 6, output_size, 1, bias
        self.rnn = _JanetLayerParams(6, hidden_size)
        self.fc_out = nn.Linear(hidden_size, output_size, bias=True)
        self._setup_delta(hidden_size, 0.0, 0.0)
        self.thx, self.thh = thx, thh

    def reset_parameters(self):
        init_gatewise(self.rnn, self.hidden_size)
        init_linear(self.fc_out, "xavier")

    def get_temporal_sparsity(self):
        out = self._sparsity(0)
        out.pop("HW_PARAM", None)          # the layer's own report (deltajanet.py:143-151) has the three ratios only
        return out
