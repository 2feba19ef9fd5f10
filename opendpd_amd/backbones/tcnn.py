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
