"""HIP-backed TCNN backbone (reference backbones/tcnn.py:5-97).

Parameters live in the same nn.Sequential layout as the reference (keys network.{0,2,4,6,8,10}.weight, network.0.bias)
with PyTorch's default Conv1d initialisation (the reference defines no reset_parameters for this class).
Kernels: csrc/tcnn.hip."""
import torch.nn as nn

from .native import NativeBackbone


class TCNN(NativeBackbone):
    backbone_name = "tcnn"

    def __init__(self, hidden_channels):
        super().__init__()
        C = hidden_channels
        self.in_channels, self.hidden_channels, self.out_channels, self.kernel_size = 6, C, 2, 5
        layers = [nn.Conv1d(6, C, kernel_size=1), nn.Hardswish()]
        for d in (1, 2, 4, 8):
            layers += [nn.Conv1d(C, C, 5, stride=1, padding=2 * d, dilation=d, groups=C, bias=False), nn.Hardswish()]
        layers += [nn.Conv1d(C, 2, kernel_size=1, bias=False)]
        self.network = nn.Sequential(*layers)
        self._finalize(C)


class NeuralTX(NativeBackbone):
    """HIP-backed NeuralTX backbone (reference backbones/neuraltx.py:5-137): complex 5-tap FIR (`conv_I`, `conv_Q`), the TCNN
    stack on the 4 features [f_I, f_Q, |f|, |f|^3] of the filtered signal, `IQ_match` (2 x 2) and identity skips of f.  Same modules,
    names, construction order and initialisation as the reference — `reset_parameters()` runs in the constructor AND once more from
    the registry (models.py:144-148), re-drawing the FIR taps and IQ_match while `network` keeps PyTorch's default init.
    Kernels: the NTX instantiation of csrc/tcnn.hip."""
    backbone_name = "neuraltx"

    def __init__(self, hidden_channels):
        super().__init__()
        C = self.hidden_channels = hidden_channels
        self.in_channels, self.out_channels, self.kernel_size, self.window_size = 4, 2, 5, 5
        self.conv_I = nn.Conv1d(1, 1, 5, bias=False, padding=2)
        self.conv_Q = nn.Conv1d(1, 1, 5, bias=False, padding=2)
        layers = [nn.Conv1d(4, C, kernel_size=1), nn.Hardswish()]
        for d in (1, 2, 4, 8):
            layers += [nn.Conv1d(C, C, 5, stride=1, padding=2 * d, dilation=d, groups=C, bias=False), nn.Hardswish()]
        layers += [nn.Conv1d(C, 2, kernel_size=1, bias=False)]
        self.network = nn.Sequential(*layers)
        self.IQ_match = nn.Linear(2, 2, bias=False)
        self.reset_parameters()
        self._finalize(C)

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.conv_I.weight, gain=0.1)
        nn.init.xavier_uniform_(self.conv_Q.weight, gain=0.1)
        nn.init.xavier_uniform_(self.IQ_match.weight, gain=1.0)     # neuraltx.py:40-55: `network` has no .weight of its own
