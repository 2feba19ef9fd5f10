"""HIP-backed generalised memory polynomial (reference backbones/gmp.py:5-50).

One parameter, `Weight` (1, M * (1 + (degree - 1) * M)), real weights on complex basis terms, xavier-uniform through
`reset_parameters()` as the reference does (gmp.py:13-16; the registry builds `GMP()` without arguments whatever hidden_size
is, models.py:26-28: memory_length 11, degree 5 -> 495 parameters).  Kernels: csrc/gmp.hip (that configuration only)."""
import torch
import torch.nn as nn

from .native import NativeBackbone


class GMP(NativeBackbone):
    backbone_name = "gmp"

    def __init__(self, memory_length=11, degree=5):
        super().__init__()
        if (memory_length, degree) != (11, 5):
            raise NotImplementedError("the HIP kernels cover the configuration the registry builds (memory_length 11, degree 5)")
        self.memory_length, self.degree = memory_length, degree
        self.W = 1 + (degree - 1) * memory_length
        self.Weight = nn.Parameter(torch.zeros(1, memory_length * self.W))      # the reference leaves it uninitialised here
        self._finalize(memory_length)

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.Weight)
