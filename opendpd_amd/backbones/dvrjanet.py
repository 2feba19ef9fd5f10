"""HIP-backed DVRJANET backbone (reference backbones/dvrjanet.py:5-112).

Parameters (names / shapes / init as the reference): cs (num_dvr_units, standard normal, dvrjanet.py:20); W_ph, W_ah:
Linear(H -> H, no bias); W_pθ, W_ax: Linear(1 -> H, no bias); W_f: Linear(H -> H); W_ccos, W_csin: Linear(2H -> H);
W_o1, W_o2: Linear(H -> 1); xavier-uniform weights, zero biases (dvrjanet.py:105-112).  Kernels: csrc/dvrjanet_s16.hip
(hidden <= 16, num_dvr_units <= 8; `odpd_model_t::bits_w` carries num_dvr_units)."""
import torch
import torch.nn as nn

from .native import NativeBackbone

MAX_HIDDEN, MAX_DVR_UNITS = 16, 8


class DVRJANET(NativeBackbone):
    backbone_name = "dvrjanet"

    def __init__(self, hidden_size, output_size=2, num_dvr_units=4, bias=True):
        super().__init__()
        if not bias or output_size != 2:
            raise NotImplementedError("dvrjanet kernels implement bias=True, I/Q output")
        if hidden_size > MAX_HIDDEN or not 1 <= num_dvr_units <= MAX_DVR_UNITS:
            raise NotImplementedError(f"dvrjanet kernels cover hidden_size <= {MAX_HIDDEN}, 1 <= num_dvr_units <= {MAX_DVR_UNITS}")
        H = self.hidden_size = hidden_size
        self.output_size, self.num_dvr_units, self.bias, self.num_layers = output_size, num_dvr_units, bias, 1
        self.W_ph = nn.Linear(H, H, bias=False)
        self.W_pθ = nn.Linear(1, H, bias=False)
        self.W_ah = nn.Linear(H, H, bias=False)
        self.W_ax = nn.Linear(1, H, bias=False)
        self.cs = nn.Parameter(torch.randn(num_dvr_units))
        self.W_f = nn.Linear(H, H, bias=True)
        self.W_ccos = nn.Linear(2 * H, H, bias=True)
        self.W_csin = nn.Linear(2 * H, H, bias=True)
        self.W_o1 = nn.Linear(H, 1, bias=True)
        self.W_o2 = nn.Linear(H, 1, bias=True)
        self._finalize(hidden_size, bits_w=num_dvr_units)

    def reset_parameters(self):
        for m in (self.W_ph, self.W_pθ, self.W_ah, self.W_ax, self.W_f, self.W_ccos, self.W_csin, self.W_o1, self.W_o2):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
