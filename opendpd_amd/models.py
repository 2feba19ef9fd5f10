"""Registry + cascade: drop-in for the reference's models.py.

`CoreModel(input_size, hidden_size, num_layers, backbone_type, window_size=None, num_dvr_units=None,
thx=0, thh=0)` — same constructor, attributes, `forward(x, h_0=None)` contract and state-dict keys
as models.py:10-160; `CascadedModel(dpd_model, pa_model)` + `freeze_pa_model()` as models.py:163-176.
Backbones on the hot path run as HIP kernels (`backbone.native` is True) inside the kernels' envelope (one layer — gru / dgru / qgru /
qgru_amp1 / lstm also two layers of <= 32 units, csrc/gru_layers2.hip, lstm_layers2.hip —, hidden
<= 32 — gru / dgru / qgru / qgru_amp1 / lstm / vdlstm / deltagru / deltagru_tcnskip / deltajanet: <= 64, csrc/*_wide.hip; pgjanet <= 32 (janet_wide.hip); tcnn, neuraltx <= 64 channels; gmp as the registry builds it; rvtdcnn fc_hid_size <= 32; dvrjanet <= 16 with <= 8 DVR units; bojanet <= 16; apnrru <= 14) and as ATen restatements (backbones/wide.py,
`native` False, with a warning) beyond it — backbones/wide.py for the hot-path names, backbones/extras.py for the SURVEY §8 f4 ones;
mcldnn: <= 16 channels.  All 18 registry names are HIP-backed inside their envelopes.  Unknown names raise ValueError (models.py:139-141).
"""
import warnings

import torch
import torch.nn as nn

from . import backbones as B
from .backbones import extras as X
from .backbones import wide as W

# names the reference registry accepts (models.py:26-141)
REFERENCE_BACKBONES = ("gmp", "gru", "dgru", "qgru", "qgru_amp1", "lstm", "vdlstm", "rvtdcnn", "apnrru", "bojanet",
                       "deltagru", "deltajanet", "pgjanet", "dvrjanet", "deltagru_tcnskip", "tcnn", "neuraltx", "mcldnn")


class CoreModel(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers, backbone_type, window_size=None, num_dvr_units=None,
                 thx=0, thh=0):
        super().__init__()
        self.output_size = 2
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.backbone_type = backbone_type
        self.thx, self.thh = thx, thh
        self.window_size, self.num_dvr_units = window_size, num_dvr_units
        self.batch_first, self.bidirectional, self.bias = True, False, True

        kw = dict(hidden_size=hidden_size, output_size=2, num_layers=num_layers, bidirectional=False,
                  batch_first=True, bias=True)
        if W.outside_envelope(backbone_type, hidden_size, num_layers):
            # beyond the kernels' hidden-size / layer-count envelope: same module through ATen (backbones/wide.py), said aloud
            W.announce(backbone_type, hidden_size, num_layers)
            self.backbone = W.build(backbone_type, input_size, hidden_size, num_layers, thx=thx, thh=thh)
        elif backbone_type == "gru":
            self.backbone = B.GRU(input_size=input_size, **kw)
        elif backbone_type == "dgru":
            self.backbone = B.DGRU(**kw)
        elif backbone_type == "qgru":
            self.backbone = B.QGRU(**kw)
        elif backbone_type == "qgru_amp1":
            self.backbone = B.QGRUAmp1(**kw)
        elif backbone_type == "lstm":
            self.backbone = B.LSTM(input_size=input_size, **kw)
        elif backbone_type == "vdlstm":
            self.backbone = B.VDLSTM(input_size=input_size, **kw)
        elif backbone_type == "deltagru":
            self.backbone = B.DeltaGRU(input_size=6, hidden_size=hidden_size, output_size=2, num_layers=num_layers, thx=thx,
                                       thh=thh, bias=True)
        elif backbone_type == "deltagru_tcnskip":
            self.backbone = B.TResDeltaGRU(input_size=6, hidden_size=hidden_size, output_size=2, num_layers=num_layers,
                                           thx=thx, thh=thh, bias=True)
        elif backbone_type == "pgjanet":
            # reference defect: models.py:109-114 passes window_size= to a ctor that has no such argument
            self.backbone = B.PGJANET(hidden_size=hidden_size, output_size=2, bias=True)
        elif backbone_type == "tcnn":
            self.backbone = B.TCNN(hidden_channels=hidden_size)
        elif backbone_type == "gmp":
            self.backbone = B.GMP()
        elif backbone_type == "rvtdcnn":
            self.backbone = B.RVTDCNN(fc_hid_size=hidden_size)
        elif backbone_type == "apnrru":
            from .backbones import apnrru as AP
            if hidden_size <= AP.MAX_HIDDEN:
                self.backbone = B.APNRRU(hidden_size=hidden_size, bias=True)
            else:
                warnings.warn(f"opendpd_amd: backbone 'apnrru' with hidden_size={hidden_size} is outside the HIP kernel's envelope "
                              f"(hidden <= {AP.MAX_HIDDEN}): running the ATen restatement (backbones/extras.py)", stacklevel=2)
                self.backbone = X.APNRRU(hidden_size=hidden_size, bias=True)
        elif backbone_type == "bojanet":
            from .backbones import bojanet as BJ
            if hidden_size <= BJ.MAX_HIDDEN:
                self.backbone = B.BOJANET(hidden_size=hidden_size, output_size=2, bias=True)
            else:
                # hidden 17, 18 (beyond 18 the reference's own forward fails, bojanet.py:41-53): outside the kernel's one unit tile
                warnings.warn(f"opendpd_amd: backbone 'bojanet' with hidden_size={hidden_size} is outside the HIP kernel's envelope "
                              f"(hidden <= {BJ.MAX_HIDDEN}): running the ATen restatement (backbones/extras.py)", stacklevel=2)
                self.backbone = X.BOJANET(hidden_size=hidden_size, output_size=2, bias=True)
        elif backbone_type == "deltajanet":
            self.backbone = B.DeltaJANET(input_size=6, hidden_size=hidden_size, output_size=2, num_layers=num_layers, thx=thx,
                                         thh=thh, bias=True)
        elif backbone_type == "dvrjanet":
            from .backbones import dvrjanet as D
            if hidden_size <= D.MAX_HIDDEN and isinstance(num_dvr_units, int) and 1 <= num_dvr_units <= D.MAX_DVR_UNITS:
                self.backbone = B.DVRJANET(hidden_size=hidden_size, output_size=2, num_dvr_units=num_dvr_units, bias=True)
            else:
                # beyond the kernel's envelope: the torch restatement through ATen, said aloud
                warnings.warn(f"opendpd_amd: backbone 'dvrjanet' with hidden_size={hidden_size}, num_dvr_units={num_dvr_units} is outside "
                              f"the HIP kernel's envelope (hidden <= {D.MAX_HIDDEN}, num_dvr_units <= {D.MAX_DVR_UNITS}): running the "
                              f"ATen restatement (backbones/extras.py)", stacklevel=2)
                self.backbone = X.DVRJANET(hidden_size=hidden_size, output_size=2, num_dvr_units=num_dvr_units, bias=True)
        elif backbone_type == "neuraltx":
            self.backbone = B.NeuralTX(hidden_channels=hidden_size)
        elif backbone_type == "mcldnn":
            from .backbones import mcldnn as MC
            if hidden_size <= MC.MAX_HIDDEN:
                self.backbone = B.MCLDNN(hidden_size=hidden_size)
            else:
                warnings.warn(f"opendpd_amd: backbone 'mcldnn' with hidden_size={hidden_size} is outside the HIP kernel's envelope "
                              f"(channels <= {MC.MAX_HIDDEN}): running the ATen restatement (backbones/extras.py)", stacklevel=2)
                self.backbone = X.MCLDNN(hidden_size=hidden_size)
        else:
            raise ValueError(f"The backbone type '{backbone_type}' is not supported. Please add your own "
                             f"backbone under ./backbones and update models.py accordingly.")
        try:  # models.py:144-148
            self.backbone.reset_parameters()
        except AttributeError:
            pass

    def forward(self, x, h_0=None):
        if not getattr(self.backbone, "native", True):
            if h_0 is None:  # models.py:154-155
                h_0 = torch.zeros(self.num_layers, x.size(0), self.hidden_size, device=x.device, dtype=x.dtype)
            # ATen restatement (outside the HIP kernels' envelope): its nn.GRU / nn.LSTM / conv layers run on ATen's own kernels, not
            # through MIOpen — whose per-shape solver search and kernel compilation on a fresh box cost minutes (285 s of the r04 GPU suite
            # were ONE such model) for layers of a few hundred parameters; the backward runs under autograd with the same setting recorded
            with torch.backends.cudnn.flags(enabled=False):
                return self.backbone(x, h_0)
        # the reference creates a zero h_0 (models.py:154-155); the kernels start from the zero state
        if h_0 is not None and bool((h_0 != 0).any()):
            raise NotImplementedError("non-zero initial hidden state is not supported by the HIP kernels")
        return self.backbone(x, None)


class CascadedModel(nn.Module):
    """y = PA(DPD(x)) with the PA frozen during DPD learning (models.py:163-176)."""

    def __init__(self, dpd_model, pa_model):
        super().__init__()
        self.dpd_model = dpd_model
        self.pa_model = pa_model

    def freeze_pa_model(self):
        for p in self.pa_model.parameters():
            p.requires_grad = False

    def forward(self, x):
        return self.pa_model(self.dpd_model(x))
