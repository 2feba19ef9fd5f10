"""Torch restatements of the SURVEY §8 f4 registry backbones (rvtdcnn, neuraltx, mcldnn, bojanet, apnrru, dvrjanet, deltajanet) for
configurations BEYOND their HIP kernels' envelopes (csrc/{rvtdcnn,tcnn,mcldnn,bojanet_s16,apnrru_s16,dvrjanet_s16,delta_s16}.hip: hidden /
channel limits, see models.py) — inside the envelopes the registry builds the kernel-backed classes of this package.

Plain restatements of what the reference modules compute (same parameter names / shapes / registration order so checkpoints
interchange, same initialisation order so a seeded construction gives the reference's state dict; `tests/test_extras_cpu.py` pins
outputs and gradients to vectors produced by running the reference).  They run through ATen on whatever device their tensors live
on — they are NOT part of the MI355X-native hot path: `CoreModel(..).backbone.native` is False for them, a warning says so at
construction, the fused optimisers refuse them and `Project.build_optimizer` gives them torch's.  Where the reference module has an
observable quirk it is kept and marked `# ref quirk`.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _polar(x):
    i, q = x[..., 0:1], x[..., 1:2]
    a2 = i * i + q * q
    return i, q, a2, torch.sqrt(a2)


def _causal_windows(x, size):
    """(B,T,C) -> (B,T,size,C): window t holds samples t-size+1 .. t, zeros before the frame start."""
    xp = F.pad(x, (0, 0, size - 1, 0))
    return xp.unfold(1, size, 1).transpose(2, 3)


def _circular_windows(x, size):
    """(B,T,C) -> (B,T,C,size): like _causal_windows but the frame's own last size-1 samples stand in front."""
    xp = torch.cat((x[:, -(size - 1):, :], x), dim=1)
    return xp.unfold(1, size, 1)


class RVTDCNN(nn.Module):
    """backbones/rvtdcnn.py:9-62 — per step a 4x5 patch [window x (I,Q,a,a^2,a^3)] -> Conv2d(1->3,k3,pad(1,0)) -> tanh ->
    Linear(36->H) -> tanh -> Linear(H->2); the window wraps around the frame."""
    native = False

    def __init__(self, window_size=4, out_channels=3, kernel_size=3, stride=1, padding=(1, 0), dilation=1, fc_hid_size=6):
        super().__init__()
        self.window_size, self.out_channels, self.fc_hid_size = window_size, out_channels, fc_hid_size
        self.fc_in_features = out_channels * 3 * window_size
        self.Conv2d = nn.Conv2d(1, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation, bias=True)
        self.fc_hid = nn.Linear(self.fc_in_features, fc_hid_size)
        self.fc_out = nn.Linear(fc_hid_size, 2)

    def forward(self, x, h_0=None):
        B, T = x.shape[0], x.shape[1]
        i, q, a2, a = _polar(x)
        feat = torch.cat((i, q, a, a2, a ** 3), dim=-1)
        win = _circular_windows(feat, self.window_size).transpose(2, 3).reshape(B * T, 1, self.window_size, 5)
        z = torch.tanh(self.Conv2d(win)).reshape(B * T, self.fc_in_features)
        return self.fc_out(torch.tanh(self.fc_hid(z))).reshape(B, T, 2)


class NeuralTX(nn.Module):
    """backbones/neuraltx.py:5-137 — complex 5-tap FIR, then the TCNN stack on [I,Q,a,a^3] of the filtered signal, plus a
    2x2 linear and identity skip of the filtered signal."""
    native = False

    def __init__(self, hidden_channels):
        super().__init__()
        C = self.hidden_channels = hidden_channels
        self.conv_I = nn.Conv1d(1, 1, 5, bias=False, padding=2)
        self.conv_Q = nn.Conv1d(1, 1, 5, bias=False, padding=2)
        layers = [nn.Conv1d(4, C, 1), nn.Hardswish()]
        for d in (1, 2, 4, 8):
            layers += [nn.Conv1d(C, C, 5, padding=2 * d, dilation=d, groups=C, bias=False), nn.Hardswish()]
        layers.append(nn.Conv1d(C, 2, 1, bias=False))
        self.network = nn.Sequential(*layers)
        self.IQ_match = nn.Linear(2, 2, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.conv_I.weight, gain=0.1)
        nn.init.xavier_uniform_(self.conv_Q.weight, gain=0.1)
        nn.init.xavier_uniform_(self.IQ_match.weight, gain=1.0)     # ref quirk: `network` keeps its default init

    def forward(self, x, h_0=None):
        # ref quirk: the reference takes an FFT over a length-1 axis first, which is the identity
        i, q = x[..., 0:1].transpose(1, 2), x[..., 1:2].transpose(1, 2)
        fi = (self.conv_I(i) - self.conv_Q(q)).transpose(1, 2)
        fq = (self.conv_Q(i) + self.conv_I(q)).transpose(1, 2)
        a = torch.sqrt(fi * fi + fq * fq)
        iq = torch.cat((fi, fq), dim=-1)
        z = self.network(torch.cat((fi, fq, a, a ** 3), dim=-1).transpose(1, 2)).transpose(1, 2)
        return z + self.IQ_match(iq) + iq


class BOJANET(nn.Module):
    """backbones/bojanet.py:5-138 — 16-tap complex FIR bank (6 filters) -> vector demodulator (|.|, |.|^2, phase) ->
    JANET cell on the envelopes -> phase re-rotation of the state -> two linear read-outs.  ATen restatement for hidden 17 and 18 — the
    two sizes between the HIP kernel's one unit tile (csrc/bojanet_s16.hip, <= 16) and the point where the reference's own phase
    re-rotation stops building (bojanet.py:41-53, > 18); retired in r03, restored in r04."""
    native = False

    def __init__(self, hidden_size, output_size=2, bias=True):
        super().__init__()
        self.hidden_size, self.output_size, self.window_size, self.num_vd_units = hidden_size, output_size, 16, 6
        H, P = hidden_size, 6
        self.fir_I = nn.Linear(16, P, bias=False)
        self.fir_Q = nn.Linear(16, P, bias=False)
        self.W_fi = nn.Linear(2 * P, H, bias=bias)
        self.W_fh = nn.Linear(H, H, bias=False)
        self.W_gi = nn.Linear(2 * P, H, bias=bias)
        self.W_gh = nn.Linear(H, H, bias=False)
        self.W_out_I = nn.Linear(H, 1, bias=bias)
        self.W_out_Q = nn.Linear(H, 1, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        for m in (self.fir_I, self.fir_Q):
            nn.init.xavier_uniform_(m.weight, gain=0.1)
        for m in (self.W_fi, self.W_gi):
            nn.init.xavier_uniform_(m.weight, gain=1.0)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        for m in (self.W_fh, self.W_gh):
            nn.init.orthogonal_(m.weight, gain=1.0)
        for m in (self.W_out_I, self.W_out_Q):
            nn.init.xavier_uniform_(m.weight, gain=1.0)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def forward(self, x, h_0=None):
        # The cell divides by the FIR outputs' magnitude (gain-0.1 taps): training amplifies rounding-level differences within tens
        # of steps, so the ATen calls below are issued in the reference's own order and grouping (per-step gate projections, the
        # same window tensor layout) — on the same device the trajectory is then bit-identical (bojanet.py:55-111).
        B, T, H, P, M = x.shape[0], x.shape[1], self.hidden_size, self.num_vd_units, self.window_size
        h = x.new_zeros(B, H) if h_0 is None else (h_0[0] if h_0.dim() == 3 else h_0)
        xp = torch.cat((torch.zeros_like(x[:, -(M - 1):, :]), x), dim=1)
        win = xp.unfold(dimension=1, size=M, step=1).transpose(2, 3).unsqueeze(2).contiguous().view(-1, T, M, x.size(2))
        fi = (self.fir_I(win[:, :, :, 0]) - self.fir_Q(win[:, :, :, 1])).contiguous().view(-1, T, P)
        fq = (self.fir_Q(win[:, :, :, 0]) + self.fir_I(win[:, :, :, 1])).contiguous().view(-1, T, P)
        mag = torch.sqrt(torch.pow(fi, 2) + torch.pow(fq, 2)) + 1e-8
        mag2 = mag ** 2
        sin, cos = fq / mag, fi / mag
        env = torch.stack([mag, mag2], dim=2).view(-1, T, 2 * P)
        hs = []
        for t in range(T):
            e = env[:, t, :]
            f = torch.sigmoid(self.W_fi(e) + self.W_fh(h))
            g = torch.tanh(self.W_gi(e) + self.W_gh(h))
            h = f * h + (1 - f) * g
            hs.append(h)
        hs = torch.stack(hs, dim=1).view(-1, T, H)
        if P >= H:
            cos, sin = cos[:, :, :H], sin[:, :, :H]
        elif H <= 2 * P:
            cos, sin = torch.cat([cos, cos[:, :, :H - P]], dim=-1), torch.cat([sin, sin[:, :, :H - P]], dim=-1)
        else:
            cos = torch.cat([cos, cos, cos[:, :, :H - 2 * P]], dim=-1)
            sin = torch.cat([sin, sin, sin[:, :, :H - 2 * P]], dim=-1)
        i_rot, q_rot = hs * cos, hs * sin
        out_i = self.W_out_I(i_rot) - self.W_out_Q(q_rot)
        out_q = self.W_out_Q(q_rot) + self.W_out_I(i_rot)       # ref quirk: both outputs mix the two read-outs
        return torch.cat([out_i, out_q], dim=-1)


class MCLDNN(nn.Module):
    """backbones/mcldnn.py:9-134 — per step a 5x5 patch [(I,Q,a,a^2,a^3) x memory]: Conv2d branch and grouped Conv1d branch,
    merged by a second Conv2d, then nn.LSTM(5C->8) over time and two Linear layers."""
    native = False

    def __init__(self, hidden_size=8):
        super().__init__()
        self.memory_length, self.input_height, self.channels = 5, 5, hidden_size
        C = hidden_size
        self.conv2d_1 = nn.Conv2d(1, C, 3, padding=1)
        self.conv1d = nn.Conv1d(5, 5 * C, 3, padding=1, groups=5)
        self.conv2d_2 = nn.Conv2d(10, 1, 3, padding=1)
        self.lstm = nn.LSTM(input_size=C * 5, hidden_size=8, num_layers=1, batch_first=True)
        self.fc_out = nn.Linear(8, 16)
        self.fc_out_2 = nn.Linear(16, 2)
        self.reset_parameters()

    def reset_parameters(self):
        for name, p in self.named_parameters():
            if "weight" in name:
                nn.init.xavier_uniform_(p)
            elif "bias" in name:
                nn.init.constant_(p, 0)

    def forward(self, x, h_0=None):
        B, T, C = x.shape[0], x.shape[1], self.channels
        i, q, a2, a = _polar(x)
        patch = _circular_windows(torch.cat((i, q, a, a2, a ** 3), dim=-1), 5).reshape(B * T, 1, 5, 5)
        p2 = self.conv2d_1(patch)                                          # (N,C,5,5)
        p1 = self.conv1d(patch.squeeze(1)).reshape(B * T, C, 5, 5)          # ref quirk: (5C) re-read as (C,5)
        z = self.conv2d_2(torch.cat((p2, p1), dim=2).transpose(1, 2)).reshape(B, T, 5 * C)
        z, _ = self.lstm(z)
        return self.fc_out_2(self.fc_out(z))


class RRU(nn.Module):
    """backbones/apnrru.py:5-33 — v = sigmoid(C h) + Z tanh(W_h tanh(W_u [x, h]))."""

    def __init__(self, hidden_size, window_size, bias=True):
        super().__init__()
        n = 2 * hidden_size + 3
        self.W_u = nn.Linear(n + 3 * 2 + 2, 16, bias=bias)
        self.W_h = nn.Linear(16, n, bias=bias)
        self.C = nn.Parameter(torch.rand(1))
        self.Z = nn.Parameter(torch.zeros(1, n))

    def forward(self, x, h_prev, h_A_prev):
        h = torch.cat((h_prev, h_A_prev), dim=-1)
        v = torch.tanh(self.W_h(torch.tanh(self.W_u(torch.cat((x, h), dim=-1)))))
        return torch.sigmoid(self.C * h) + self.Z * v


class APNRRU(nn.Module):
    """backbones/apnrru.py:36-152 — phase-normalised recurrent unit: the FIR outputs and the complex state are rotated by
    the conjugate phase of the current sample before the cell and rotated back after it."""
    native = False

    def __init__(self, hidden_size, bias=True):
        super().__init__()
        self.hidden_size, self.hidden_size_A, self.window_size, self.num_fir_filters = hidden_size, 3, 16, 3
        self.fir_I = nn.Linear(16, 3, bias=False)
        self.fir_Q = nn.Linear(16, 3, bias=False)
        self.rru = RRU(hidden_size, 16, bias)
        self.output_layer_I = nn.Linear(hidden_size, 1, bias=False)
        self.output_layer_Q = nn.Linear(hidden_size, 1, bias=False)

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.fir_I.weight)
        nn.init.xavier_uniform_(self.fir_Q.weight)
        for m in (self.rru.W_u, self.rru.W_h):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        # ref quirk: the reference then touches a non-existent `output_layer` (AttributeError swallowed by CoreModel), so
        # the two read-out layers keep their default initialisation
        raise AttributeError("APNRRU has no attribute 'output_layer'")

    def forward(self, x, h_0=None):
        B, T, H = x.shape[0], x.shape[1], self.hidden_size
        win = _causal_windows(x, self.window_size)
        wi, wq = win[..., 0], win[..., 1]
        mag = torch.sqrt(x[..., 0] ** 2 + x[..., 1] ** 2)
        rr, ri = x[..., 0] / mag, -x[..., 1] / mag                                  # r = conj(x)/|x|
        fi = torch.cat((self.fir_I(wi) - self.fir_Q(wq), x[..., 0:1]), dim=-1)      # (B,T,4)
        fq = torch.cat((self.fir_Q(wi) + self.fir_I(wq), x[..., 1:2]), dim=-1)
        ni = rr.unsqueeze(-1) * fi - ri.unsqueeze(-1) * fq
        nq = ri.unsqueeze(-1) * fi + rr.unsqueeze(-1) * fq
        feats = torch.stack((ni, nq), dim=-1).reshape(B, T, 8)
        hI, hQ, hA = x.new_zeros(B, H), x.new_zeros(B, H), x.new_zeros(B, 3)
        out = []
        for t in range(T):
            c, s = rr[:, t:t + 1], ri[:, t:t + 1]
            hI, hQ = hI * c - hQ * s, hI * s + hQ * c                  # state into the normalised frame
            v = self.rru(feats[:, t], torch.cat((hI, hQ), dim=-1), hA)
            vI, vQ, hA = v[:, :H], v[:, H:2 * H], v[:, 2 * H:]
            hI, hQ = vI * c + vQ * s, vQ * c - vI * s                  # and back: multiply by conj(r)
            yi, yq = self.output_layer_I(hI), self.output_layer_Q(hQ)
            out.append(torch.cat((yi - yq, yq + yi), dim=-1))          # ref quirk as in BOJANET
        return torch.stack(out, dim=1)


class DVRJANET(nn.Module):
    """backbones/dvrjanet.py:5-112 — phase and magnitude recurrent filters, decomposed-vector-rotation nonlinearity on the
    magnitude branch, JANET-style update of an I state and a Q state."""
    native = False

    def __init__(self, hidden_size, output_size=2, num_dvr_units=4, bias=True):
        super().__init__()
        H = self.hidden_size = hidden_size
        self.output_size, self.num_dvr_units = output_size, num_dvr_units
        self.W_ph = nn.Linear(H, H, bias=False)
        self.W_pθ = nn.Linear(1, H, bias=False)
        self.W_ah = nn.Linear(H, H, bias=False)
        self.W_ax = nn.Linear(1, H, bias=False)
        self.cs = nn.Parameter(torch.randn(num_dvr_units))
        self.W_f = nn.Linear(H, H, bias=bias)
        self.W_ccos = nn.Linear(2 * H, H, bias=bias)
        self.W_csin = nn.Linear(2 * H, H, bias=bias)
        self.W_o1 = nn.Linear(H, 1, bias=bias)
        self.W_o2 = nn.Linear(H, 1, bias=bias)

    def reset_parameters(self):
        for m in (self.W_ph, self.W_pθ, self.W_ah, self.W_ax, self.W_f, self.W_ccos, self.W_csin, self.W_o1, self.W_o2):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def forward(self, x, h_0=None):
        B, T, H, K = x.shape[0], x.shape[1], self.hidden_size, self.num_dvr_units
        mag = torch.sqrt(x[..., 0:1] ** 2 + x[..., 1:2] ** 2)
        th = torch.atan2(x[..., 1:2], x[..., 0:1])
        pin, ain = self.W_pθ(th), self.W_ax(mag)           # input halves for all steps
        knots = torch.arange(1, K + 1, device=x.device, dtype=x.dtype) / K
        hI = x.new_zeros(B, H) if h_0 is None else h_0.squeeze(0)
        hQ = hI
        ys = []
        for t in range(T):
            hs = hI + hQ
            ph = pin[:, t] + self.W_ph(hs)
            pre = ain[:, t] + self.W_ah(hs)
            amp = sum(torch.abs(pre - knots[k]) * self.cs[k] for k in range(K))
            f = torch.sigmoid(self.W_f(hs))
            gI = torch.tanh(self.W_ccos(torch.cat((hI, amp * torch.cos(ph)), dim=-1)))
            gQ = torch.tanh(self.W_csin(torch.cat((hQ, amp * torch.sin(ph)), dim=-1)))
            hI = f * hI + (1 - f) * gI
            hQ = f * hQ + (1 - f) * gQ
            ys.append(torch.cat((self.W_o1(hI), self.W_o2(hQ)), dim=-1))
        return torch.stack(ys, dim=1).reshape(B, T, self.output_size)


class DeltaJANETLayer(nn.Module):
    """backbones/deltajanet.py:67-274 — JANET cell driven by accumulated input / state deltas (pre-activation memory `dm`).
    Thresholds below which a delta is dropped; the reference's registry always builds it with both at 0."""

    def __init__(self, input_size=6, hidden_size=256, num_layers=1, thx=0.1, thh=0):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers, self.th_x, self.th_h = input_size, hidden_size, num_layers, thx, thh
        self.debug = 1
        self.set_debug(1)
        for l in range(num_layers):
            setattr(self, f"weight_ih_l{l}", nn.Parameter(torch.empty(2 * hidden_size, input_size)))
            setattr(self, f"weight_hh_l{l}", nn.Parameter(torch.empty(2 * hidden_size, hidden_size)))
            setattr(self, f"bias_ih_l{l}", nn.Parameter(torch.empty(2 * hidden_size)))
            setattr(self, f"bias_hh_l{l}", nn.Parameter(torch.empty(2 * hidden_size)))
        self.reset_parameters()

    def set_debug(self, value):
        self.debug = value
        self.statistics = {"num_dx_zeros": 0, "num_dx_numel": 0, "num_dh_zeros": 0, "num_dh_numel": 0}

    def reset_parameters(self):
        for name, p in self.named_parameters():
            if "weight" in name:
                nn.init.orthogonal_(p)
            elif "bias" in name:
                nn.init.constant_(p, 0)

    def get_temporal_sparsity(self):
        s, out = self.statistics, {}
        if self.debug and s["num_dx_numel"]:
            out["SP_T_DX"] = float(s["num_dx_zeros"] / s["num_dx_numel"])
            out["SP_T_DH"] = float(s["num_dh_zeros"] / s["num_dh_numel"])
            out["SP_T_DV"] = float((s["num_dx_zeros"] + s["num_dh_zeros"]) / (s["num_dx_numel"] + s["num_dh_numel"]))
        return out

    def _layer(self, seq, l):
        w_ih, w_hh = getattr(self, f"weight_ih_l{l}"), getattr(self, f"weight_hh_l{l}")
        B, H = seq.shape[1], self.hidden_size
        dm = (getattr(self, f"bias_ih_l{l}") + getattr(self, f"bias_hh_l{l}")).unsqueeze(0).expand(B, -1)
        x_p = seq.new_zeros(B, seq.shape[2])
        h = seq.new_zeros(B, H)
        h_p = seq.new_zeros(B, H)
        outs = []
        for x in seq.unbind(0):
            dx, dh = x - x_p, h - h_p
            ax, ah = dx.abs(), dh.abs()
            dx = dx.masked_fill(ax < self.th_x, 0)
            dh = dh.masked_fill(ah < self.th_h, 0)
            if self.debug:      # counters stay on the tensors' device (no host sync per step)
                s = self.statistics
                s["num_dx_zeros"] = s["num_dx_zeros"] + (dx == 0).sum(); s["num_dx_numel"] += dx.numel()
                s["num_dh_zeros"] = s["num_dh_zeros"] + (dh == 0).sum(); s["num_dh_numel"] += dh.numel()
            x_p = torch.where(ax >= self.th_x, x, x_p)
            h_p = torch.where(ah >= self.th_h, h, h_p)
            mx = dx @ w_ih.t() + dm
            mh = dh @ w_hh.t()
            dm = mx + mh                              # both gate halves accumulate input and state deltas
            f, g = torch.sigmoid(dm[:, :H]), torch.sigmoid(dm[:, H:])      # ref quirk: the candidate gate is a sigmoid
            h = (1 - f) * g + f * h
            outs.append(h)
        return torch.stack(outs)

    def forward(self, x, *unused):
        seq = x.transpose(0, 1)
        for l in range(self.num_layers):
            seq = self._layer(seq, l)
        return seq.transpose(0, 1)


class DeltaJANET(nn.Module):
    """backbones/deltajanet.py:11-64 — [I,Q,a,a^3,sin,cos] -> DeltaJANETLayer -> Linear(H->2)."""
    native = False

    def __init__(self, input_size, hidden_size, output_size, num_layers, thx=0, thh=0, bias=True):
        super().__init__()
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, input_size, output_size, num_layers
        self.thx, self.thh = thx, thh
        self.rnn = DeltaJANETLayer(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, thx=0, thh=0)  # ref quirk: thresholds dropped
        self.fc_out = nn.Linear(hidden_size, output_size, bias=True)

    def reset_parameters(self):
        H = self.hidden_size
        for name, p in self.rnn.named_parameters():
            gates = p.shape[0] // H
            if "bias" in name:
                nn.init.constant_(p, 0)
            if "weight" in name:
                for g in range(gates):
                    nn.init.orthogonal_(p[g * H:(g + 1) * H, :])
            if "weight_ih_l0" in name:
                for g in range(gates):
                    nn.init.xavier_uniform_(p[g * H:(g + 1) * H, :])
        nn.init.xavier_uniform_(self.fc_out.weight)
        nn.init.constant_(self.fc_out.bias, 0)

    def set_debug(self, value):
        self.rnn.set_debug(value)

    def get_temporal_sparsity(self):
        return self.rnn.get_temporal_sparsity()

    def forward(self, x, h_0=None):
        i, q, a2, a = _polar(x)
        return self.fc_out(self.rnn(torch.cat((i, q, a, a ** 3, q / a, i / a), dim=-1)))
