"""HIP-backed APNRRU backbone (reference backbones/apnrru.py:5-152).

Parameters (names / shapes / init as the reference): fir_I, fir_Q: Linear(16 -> 3, no bias); rru.C (1, uniform [0, 1)), rru.Z (1, n)
zeros, rru.W_u: Linear(8 + n -> 16), rru.W_h: Linear(16 -> n) with n = 2 hidden + 3; output_layer_I / _Q: Linear(hidden -> 1, no bias).
The reference's reset_parameters() (apnrru.py:134-152) xavier-initialises the FIR banks and the two cell layers (zero biases), then
touches a non-existent `output_layer` — the AttributeError is swallowed by the registry (models.py:144-148), so the two read-outs keep
their default initialisation; the same happens here.  Kernels: csrc/apnrru_s16.hip (hidden <= 14: the state of 2 hidden + 3 values
fills two 16-slot tiles)."""
import torch
import torch.nn as nn

from .native import NativeBackbone

MAX_HIDDEN = 14


class _RRUParams(nn.Module):
    """parameter holder with the reference's names and registration order (apnrru.py:5-19): C, Z, then W_u, W_h"""

    def __init__(self, hidden_size):
        super().__init__()
        n = 2 * hidden_size + 3
        self.W_u = nn.Linear(n + 3 * 2 + 2, 16, bias=True)
        self.W_h = nn.Linear(16, n, bias=True)
        self.C = nn.Parameter(torch.rand(1))
        self.Z = nn.Parameter(torch.zeros(1, n))


class APNRRU(NativeBackbone):
    backbone_name = "apnrru"

    def __init__(self, hidden_size, bias=True):
        super().__init__()
        if not bias:
            raise NotImplementedError("apnrru kernels implement bias=True")
        if hidden_size > MAX_HIDDEN:
            raise NotImplementedError(f"apnrru kernels cover hidden_size <= {MAX_HIDDEN}")
        self.hidden_size, self.hidden_size_A, self.window_size, self.num_fir_filters, self.hidden_node = hidden_size, 3, 16, 3, 16
        self.output_size, self.num_layers = 2, 1
        self.fir_I = nn.Linear(16, 3, bias=False)
        self.fir_Q = nn.Linear(16, 3, bias=False)
        self.rru = _RRUParams(hidden_size)
        self.output_layer_I = nn.Linear(hidden_size, 1, bias=False)
        self.output_layer_Q = nn.Linear(hidden_size, 1, bias=False)
        self._finalize(hidden_size)

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.fir_I.weight)
        nn.init.xavier_uniform_(self.fir_Q.weight)
        for m in (self.rru.W_u, self.rru.W_h):
            nn.init.xavier_uniform_(m.weight)
            nn.init.constant_(m.bias, 0)
        # reference quirk (apnrru.py:149-152): the read-outs keep their default initialisation
        raise AttributeError("APNRRU has no attribute 'output_layer'")
