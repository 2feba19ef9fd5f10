"""ATen restatements of the kernel-backed backbones for configurations OUTSIDE the HIP kernels' envelope.

The HIP kernels cover what the reference's scripts and BASELINE configs use: one recurrent layer, hidden size <= 32
(pgjanet, QAT: <= 16; tcnn, neuraltx: <= 64 channels; rvtdcnn: fc_hid_size <= 32).  The reference's registry accepts any `hidden_size` / `num_layers`
(models.py:11), so `CoreModel` builds one of the modules below for a configuration beyond those limits: same parameter
names / shapes / initialisation and the same arithmetic as the reference class, executed by PyTorch-ROCm's own GPU
operators (MIOpen RNN, ATen convolutions) — `native` is False, a warning says so once per configuration, the fused
optimiser declines the model (project.py falls back to torch.optim) and nothing here is used inside the envelope.
Arithmetic follows the same reference lines as the kernels (gru.py:39-48, dgru.py:59-74, qgru.py:59-71,
qgru_amp1.py:61-76, lstm.py:39-48, vdlstm.py:58-82, deltagru.py:59-77 + 150-264, deltagru_tcnskip.py:87-103 + 195-293,
pgjanet.py:33-77, tcnn.py:82-96); tests/test_wide_cpu.py checks every class against the C oracle at hidden sizes the
kernels do not reach.
"""
import warnings

import torch
import torch.nn as nn

from .native import init_gatewise, init_linear

# largest hidden size (tcnn: channel count) the HIP kernels of a registry name run, single layer
KERNEL_HIDDEN_LIMIT = {"gru": 64, "dgru": 64, "qgru": 64, "qgru_amp1": 64, "lstm": 64, "vdlstm": 64, "deltagru": 64,
                       "deltagru_tcnskip": 64, "pgjanet": 32, "tcnn": 64, "rvtdcnn": 32, "neuraltx": 64, "deltajanet": 64}
TWO_LAYER_KERNELS = ("gru", "dgru", "qgru", "qgru_amp1", "lstm")      # two stacked recurrent layers of <= 32 units run on kernels
_warned = set()


def outside_envelope(backbone_type, hidden_size, num_layers):
    lim = KERNEL_HIDDEN_LIMIT.get(backbone_type)
    if lim is None:
        return False
    if num_layers == 2 and backbone_type in TWO_LAYER_KERNELS and hidden_size <= 32:      # csrc/gru_layers2.hip, lstm_layers2.hip
        return False
    return hidden_size > lim or (num_layers != 1 and backbone_type not in ("pgjanet", "tcnn", "rvtdcnn", "neuraltx"))


def announce(backbone_type, hidden_size, num_layers):
    key = (backbone_type, hidden_size, num_layers)
    if key not in _warned:
        _warned.add(key)
        warnings.warn(f"opendpd_amd: backbone '{backbone_type}' with hidden_size={hidden_size}, num_layers={num_layers} is outside "
                      f"the HIP kernels' envelope (hidden <= {KERNEL_HIDDEN_LIMIT[backbone_type]}, one layer): running the ATen "
                      f"restatement (backbones/wide.py), not the hand-written kernels", stacklevel=3)


def _amp(x):
    return torch.sqrt(x[..., 0] * x[..., 0] + x[..., 1] * x[..., 1])


def _feat_polar6(x):
    """[I, Q, |x|, |x|^3, sin, cos] (dgru.py:61-68, deltagru.py:61-73, tcnn.py:84-91)"""
    a = _amp(x)
    return torch.stack((x[..., 0], x[..., 1], a, a * a * a, x[..., 1] / a, x[..., 0] / a), dim=-1)


class _Wide(nn.Module):
    native = False


class WideGRU(_Wide):
    def __init__(self, input_size, hidden_size, output_size, num_layers, bidirectional=False, batch_first=True, bias=True):
        super().__init__()
        if bidirectional:
            raise NotImplementedError("bidirectional recurrences are not part of the reference's registry (models.py:24)")
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, input_size, output_size, num_layers
        self.rnn = nn.GRU(input_size, hidden_size, num_layers, batch_first=True, bias=bias)
        self.fc_out = nn.Linear(hidden_size, output_size, bias=True)

    def reset_parameters(self):
        init_gatewise(self.rnn, self.hidden_size)
        init_linear(self.fc_out, "xavier")

    def features(self, x):
        return x

    def forward(self, x, h_0):
        out, _ = self.rnn(self.features(x), h_0)
        return self.fc_out(out)


class WideQGRU(WideGRU):
    """float path of qgru.py: features [I, Q, |x|^2, |x|^4]"""

    def __init__(self, hidden_size, output_size, num_layers, bidirectional=False, batch_first=True, bias=True):
        super().__init__(4, hidden_size, output_size, num_layers, bidirectional, batch_first, bias)

    def features(self, x):
        a2 = x[..., 0] * x[..., 0] + x[..., 1] * x[..., 1]
        return torch.stack((x[..., 0], x[..., 1], a2, a2 * a2), dim=-1)


class WideQGRUAmp1(WideQGRU):
    """float path of qgru_amp1.py: features [I, Q, |x|, |x|^3]"""

    def features(self, x):
        a = _amp(x)
        return torch.stack((x[..., 0], x[..., 1], a, a * a * a), dim=-1)


class WideDGRU(_Wide):
    def __init__(self, hidden_size, output_size, num_layers, bidirectional=False, batch_first=True, bias=True):
        super().__init__()
        if bidirectional:
            raise NotImplementedError("bidirectional recurrences are not part of the reference's registry (models.py:24)")
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, 6, output_size, num_layers
        self.rnn = nn.GRU(6, hidden_size, num_layers, batch_first=True, bias=bias)
        self.fc_out = nn.Linear(hidden_size + 6, output_size, bias=True)      # registered before fc_hid, as dgru.py:26-31
        self.fc_hid = nn.Linear(hidden_size, hidden_size, bias=True)

    def reset_parameters(self):
        init_gatewise(self.rnn, self.hidden_size)
        init_linear(self.fc_out, "xavier")
        init_linear(self.fc_hid, "kaiming")

    def forward(self, x, h_0):
        f = _feat_polar6(x)
        out, _ = self.rnn(f, h_0)
        return self.fc_out(torch.cat((torch.relu(self.fc_hid(out)), f), dim=-1))


class WideLSTM(_Wide):
    def __init__(self, input_size, hidden_size, output_size, num_layers, bidirectional=False, batch_first=True, bias=True):
        super().__init__()
        if bidirectional:
            raise NotImplementedError("bidirectional recurrences are not part of the reference's registry (models.py:24)")
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, input_size, output_size, num_layers
        self.rnn = nn.LSTM(input_size, hidden_size, num_layers, batch_first=True, bias=bias)
        self.fc_out = nn.Linear(hidden_size, output_size, bias=True)

    def reset_parameters(self):
        init_gatewise(self.rnn, self.hidden_size)
        init_linear(self.fc_out, "xavier")

    def forward(self, x, h_0):
        out, _ = self.rnn(x, (h_0, h_0))       # lstm.py:46: h_0 serves as both initial h and c
        return self.fc_out(out)


class WideVDLSTM(_Wide):
    def __init__(self, input_size, hidden_size, output_size, num_layers, window_length=4, stride=1, bidirectional=False,
                 batch_first=True, bias=True):
        super().__init__()
        if bidirectional or window_length != 4 or stride != 1:
            raise NotImplementedError("vdlstm: unidirectional, window_length=4, stride=1 (the reference's defaults)")
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, window_length, output_size, num_layers
        self.window_length, self.stride, self.pad_size = window_length, stride, window_length - 1
        self.rnn = nn.LSTM(window_length, hidden_size, num_layers, batch_first=True, bias=bias)
        self.fc_lambda_1 = nn.Linear(hidden_size, window_length, bias=True)
        self.fc_lambda_2 = nn.Linear(hidden_size, window_length, bias=True)
        self.fc_out = nn.Linear(2 * window_length, 2, bias=True)

    def reset_parameters(self):
        init_gatewise(self.rnn, self.hidden_size)
        for lin in (self.fc_lambda_1, self.fc_lambda_2, self.fc_out):
            init_linear(lin, "xavier")

    def _windows(self, v):
        """(B,T) -> (B,T,4): window t = samples t-3..t of the frame, wrapping around its end (vdlstm.py:66-74)"""
        return torch.cat((v[:, -self.pad_size:], v), dim=1).unfold(1, self.window_length, self.stride)

    def forward(self, x, h_0):
        a = _amp(x)
        aw = self._windows(a)
        cw, sw = self._windows(x[..., 0]) / aw, self._windows(x[..., 1]) / aw
        out, _ = self.rnn(aw)                   # the h_0 argument is ignored (vdlstm.py:77)
        return self.fc_out(torch.cat((self.fc_lambda_1(out) * cw, self.fc_lambda_2(out) * sw), dim=-1))


class _DeltaCounters:
    def __init__(self):
        self.reset()

    def reset(self):
        self.v = [0.0, 0.0, 0.0, 0.0]

    def as_dict(self):
        return {"num_dx_zeros": self.v[0], "num_dx_numel": self.v[1], "num_dh_zeros": self.v[2], "num_dh_numel": self.v[3]}


class _WideDelta(_Wide):
    """time loop of the delta-network GRU (deltagru.py:150-264): thresholded input / state deltas drive accumulators"""

    def _init_delta(self, hidden_size, num_layers, thx, thh):
        if num_layers != 1:
            raise NotImplementedError("delta backbones: one layer (deltagru_tcnskip.py shares one x2h across layers; the "
                                      "reference scripts use one)")
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, 6, 2, 1
        self.thx, self.thh, self.debug = thx, thh, 1
        self._ctr = _DeltaCounters()

    def set_debug(self, value):
        self.debug = value
        self._ctr.reset()

    @property
    def statistics(self):
        return self._ctr.as_dict()

    def _weights(self):
        raise NotImplementedError

    def _scan(self, f):
        B, T, H = f.shape[0], f.shape[1], self.hidden_size
        w_ih, w_hh, dm0, dmnh0 = self._weights()
        xp = f.new_zeros(B, 6)
        h, hp = f.new_zeros(B, H), f.new_zeros(B, H)
        dm, dmnh = dm0.expand(B, -1), dmnh0.expand(B, -1)
        outs = []
        for t in range(T):
            dx, dh = f[:, t] - xp, h - hp
            keep_x, keep_h = ~(dx.abs() < self.thx), ~(dh.abs() < self.thh)        # masked_fill(|d| < th, 0)
            dxm, dhm = dx * keep_x, dh * keep_h
            xp = torch.where(dx.abs() >= self.thx, f[:, t], xp)
            hp = torch.where(dh.abs() >= self.thh, h, hp)
            if self.debug:
                self._ctr.v[0] += float((dxm == 0).sum()); self._ctr.v[1] += dxm.numel()
                self._ctr.v[2] += float((dhm == 0).sum()); self._ctr.v[3] += dhm.numel()
            mh = dhm @ w_hh.t()
            dm = dm + dxm @ w_ih.t() + torch.cat((mh[:, :2 * H], torch.zeros_like(mh[:, 2 * H:])), dim=1)
            dmnh = dmnh + mh[:, 2 * H:]
            r, z = torch.sigmoid(dm[:, :H]), torch.sigmoid(dm[:, H:2 * H])
            n = torch.tanh(dm[:, 2 * H:] + r * dmnh)
            h = (1 - z) * n + z * h
            outs.append(h)
        return torch.stack(outs, dim=1)

    def _sparsity(self, fc_numel):
        st, out = self._ctr.as_dict(), {}
        if self.debug and st["num_dx_numel"] > 0:
            rnn_w = sum(p.numel() for n, p in self.rnn.named_parameters() if "weight" in n)
            rnn_b = sum(p.numel() for n, p in self.rnn.named_parameters() if "bias" in n)
            tz, tn = st["num_dx_zeros"] + st["num_dh_zeros"], st["num_dx_numel"] + st["num_dh_numel"]
            out = {"SP_T_DX": float(st["num_dx_zeros"] / st["num_dx_numel"]), "SP_T_DH": float(st["num_dh_zeros"] / st["num_dh_numel"]),
                   "SP_T_DV": float(tz / tn), "HW_PARAM": float(fc_numel + rnn_w * (1 - float(tz / tn)) + rnn_b)}
        return out


class WideDeltaGRU(_WideDelta):
    def __init__(self, input_size, hidden_size, output_size, num_layers, thx=0, thh=0, bias=True):
        super().__init__()
        from .deltagru import _GruLayerParams
        self._init_delta(hidden_size, num_layers, thx, thh)
        self.rnn = _GruLayerParams(6, hidden_size)
        self.fc_out = nn.Linear(hidden_size, output_size, bias=True)

    def reset_parameters(self):
        init_gatewise(self.rnn, self.hidden_size)
        init_linear(self.fc_out, "xavier")

    def _weights(self):
        H, r = self.hidden_size, self.rnn
        b = r.bias_ih_l0 + torch.cat((r.bias_hh_l0[:2 * H], torch.zeros_like(r.bias_hh_l0[2 * H:])))   # deltagru.py:165-170
        return r.weight_ih_l0, r.weight_hh_l0, b.unsqueeze(0), r.bias_hh_l0[2 * H:].unsqueeze(0)

    def forward(self, x, h_0=None):
        return self.fc_out(self._scan(_feat_polar6(x)))

    def get_temporal_sparsity(self):
        return self._sparsity(sum(p.numel() for p in self.fc_out.parameters()))


class WideTResDeltaGRU(_WideDelta):
    def __init__(self, input_size, hidden_size, output_size, num_layers, thx=0, thh=0, bias=True):
        super().__init__()
        from .deltagru import _TresLayerParams
        self._init_delta(hidden_size, num_layers, thx, thh)
        self.rnn = _TresLayerParams(6, hidden_size)
        self.fc_out = nn.Linear(hidden_size, output_size, bias=False)
        self.tcn = nn.Sequential(nn.Conv1d(2, 3, kernel_size=3, padding=16, stride=1, dilation=16, bias=False), nn.Hardswish(),
                                 nn.Conv1d(3, 2, kernel_size=1, padding=0, stride=1, dilation=1, bias=False), nn.Hardswish())

    def reset_parameters(self):
        for name, p in self.tcn.named_parameters():
            if "weight" in name:
                nn.init.xavier_uniform_(p)
        init_gatewise(self.rnn, self.hidden_size, xavier_suffix="x2h.weight")
        init_linear(self.fc_out, "xavier")

    def _weights(self):
        H = self.hidden_size
        z = self.rnn.x2h.weight.new_zeros
        return self.rnn.x2h.weight, self.rnn.h2h.weight, z(1, 3 * H), z(1, H)      # bias-free: accumulators start at 0

    def forward(self, x, h_0=None):
        a = _amp(x)
        nxt = torch.roll(x, -1, dims=1)                                            # the last step sees the first sample
        f = torch.stack((x[..., 0], x[..., 1], a, a * a * a, nxt[..., 0], nxt[..., 1]), dim=-1)
        skip = self.tcn(x.transpose(1, 2)).transpose(1, 2)
        return self.fc_out(self._scan(f)) + skip

    def get_temporal_sparsity(self):
        return self._sparsity(sum(p.numel() for p in self.fc_out.parameters()) + sum(p.numel() for p in self.tcn.parameters()))


class WidePGJANET(_Wide):
    def __init__(self, hidden_size, output_size, bias=True):
        super().__init__()
        self.hidden_size, self.output_size, self.bias, self.num_layers = hidden_size, output_size, bias, 1
        H = hidden_size
        self.W_a, self.W_p1, self.W_p2 = (nn.Linear(H + 1, H, bias=bias) for _ in range(3))
        self.W_f, self.W_g = nn.Linear(2 * H, H, bias=bias), nn.Linear(2 * H, H, bias=bias)
        self.W_o = nn.Linear(H, output_size, bias=bias)

    def reset_parameters(self):
        for m in (self.W_a, self.W_p1, self.W_p2, self.W_f, self.W_g, self.W_o):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def forward(self, x, h_0=None):
        amp = _amp(x).unsqueeze(-1)
        th = torch.atan2(x[..., 1], x[..., 0])
        ct, st = torch.cos(th).unsqueeze(-1), torch.sin(th).unsqueeze(-1)
        h = x.new_zeros(x.shape[0], self.hidden_size)
        outs = []
        for t in range(x.shape[1]):
            a = torch.tanh(self.W_a(torch.cat((h, amp[:, t]), dim=1)))
            p1 = torch.tanh(self.W_p1(torch.cat((h, ct[:, t]), dim=1)))
            p2 = torch.tanh(self.W_p2(torch.cat((h, st[:, t]), dim=1)))
            u = a * p1 * p2 * (1 - a) * (1 - p1) * (1 - p2)
            hu = torch.cat((h, u), dim=1)
            f, g = torch.sigmoid(self.W_f(hu)), torch.tanh(self.W_g(hu))
            h = f * h + (1 - f) * g
            outs.append(self.W_o(h))
        return torch.stack(outs, dim=1)


class WideTCNN(_Wide):
    def __init__(self, hidden_channels):
        super().__init__()
        C = hidden_channels
        self.in_channels, self.hidden_channels, self.out_channels, self.kernel_size = 6, C, 2, 5
        layers = [nn.Conv1d(6, C, kernel_size=1), nn.Hardswish()]
        for d in (1, 2, 4, 8):
            layers += [nn.Conv1d(C, C, 5, stride=1, padding=2 * d, dilation=d, groups=C, bias=False), nn.Hardswish()]
        layers += [nn.Conv1d(C, 2, kernel_size=1, bias=False)]
        self.network = nn.Sequential(*layers)

    def forward(self, x, h_0=None):
        return self.network(_feat_polar6(x).transpose(1, 2)).transpose(1, 2) + x


def build(backbone_type, input_size, hidden_size, num_layers, thx=0, thh=0):
    """The wide restatement of a kernel-backed registry name (models.py:26-141 constructor arguments)."""
    kw = dict(hidden_size=hidden_size, output_size=2, num_layers=num_layers, bidirectional=False, batch_first=True, bias=True)
    if backbone_type == "gru":
        return WideGRU(input_size=input_size, **kw)
    if backbone_type == "dgru":
        return WideDGRU(**kw)
    if backbone_type == "qgru":
        return WideQGRU(**kw)
    if backbone_type == "qgru_amp1":
        return WideQGRUAmp1(**kw)
    if backbone_type == "lstm":
        return WideLSTM(input_size=input_size, **kw)
    if backbone_type == "vdlstm":
        return WideVDLSTM(input_size=input_size, **kw)
    if backbone_type == "deltagru":
        return WideDeltaGRU(6, hidden_size, 2, num_layers, thx=thx, thh=thh, bias=True)
    if backbone_type == "deltagru_tcnskip":
        return WideTResDeltaGRU(6, hidden_size, 2, num_layers, thx=thx, thh=thh, bias=True)
    if backbone_type == "pgjanet":
        return WidePGJANET(hidden_size=hidden_size, output_size=2, bias=True)
    if backbone_type == "tcnn":
        return WideTCNN(hidden_channels=hidden_size)
    if backbone_type == "rvtdcnn":
        from .extras import RVTDCNN         # the ATen restatement that served every size before csrc/rvtdcnn.hip
        return RVTDCNN(fc_hid_size=hidden_size)
    if backbone_type == "neuraltx":
        from .extras import NeuralTX
        return NeuralTX(hidden_channels=hidden_size)
    if backbone_type == "deltajanet":
        from .extras import DeltaJANET
        return DeltaJANET(input_size=6, hidden_size=hidden_size, output_size=2, num_layers=num_layers, thx=thx, thh=thh, bias=True)
    raise ValueError(backbone_type)
