"""HIP-backed BOJANET backbone (reference backbones/bojanet.py:5-138).

Parameters (names / shapes / init as the reference): fir_I, fir_Q: Linear(16 -> 6, no bias, xavier-uniform gain 0.1); W_fi, W_gi:
Linear(12 -> H) xavier-uniform + zero bias; W_fh, W_gh: Linear(H -> H, no bias) orthogonal; W_out_I, W_out_Q: Linear(H -> 1)
xavier-uniform + zero bias (bojanet.py:108-134; the constructor initialises once, the registry's reset_parameters() a second time —
both draws are kept so that a seeded construction consumes the generator as the reference does).  Kernels: csrc/bojanet_s16.hip
(hidden <= 16; the reference's own phase re-rotation stops at hidden 18)."""
import torch.nn as nn

from .native import NativeBackbone

MAX_HIDDEN = 16


class BOJANET(NativeBackbone):
    backbone_name = "bojanet"

    def __init__(self, hidden_size, output_size=2, bias=True):
        super().__init__()
        if not bias or output_size != 2:
            raise NotImplementedError("bojanet kernels implement bias=True, I/Q output")
        if hidden_size > MAX_HIDDEN:
            raise NotImplementedError(f"bojanet kernels cover hidden_size <= {MAX_HIDDEN}")
        H = self.hidden_size = hidden_size
        self.output_size, self.window_size, self.num_vd_units, self.bias, self.num_layers = output_size, 16, 6, bias, 1
        self.fir_I = nn.Linear(16, 6, bias=False)
        self.fir_Q = nn.Linear(16, 6, bias=False)
        self.W_fi = nn.Linear(12, H, bias=True)
        self.W_fh = nn.Linear(H, H, bias=False)
        self.W_gi = nn.Linear(12, H, bias=True)
        self.W_gh = nn.Linear(H, H, bias=False)
        self.W_out_I = nn.Linear(H, 1, bias=True)
        self.W_out_Q = nn.Linear(H, 1, bias=True)
        self.reset_parameters()
        self._finalize(hidden_size)

    def reset_parameters(self):
        for m in (self.fir_I, self.fir_Q):
            nn.init.xavier_uniform_(m.weight, gain=0.1)
        for m in (self.W_fi, self.W_gi):
            nn.init.xavier_uniform_(m.weight, gain=1.0)
            nn.init.constant_(m.bias, 0)
        for m in (self.W_fh, self.W_gh):
            nn.init.orthogonal_(m.weight, gain=1.0)
        for m in (self.W_out_I, self.W_out_Q):
            nn.init.xavier_uniform_(m.weight, gain=1.0)
            nn.init.constant_(m.bias, 0)
