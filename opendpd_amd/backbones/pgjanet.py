"""HIP-backed PGJANET backbone (reference backbones/pgjanet.py:5-84).

Parameters (names / shapes / init as the reference): W_a, W_p1, W_p2: Linear(H+1 -> H); W_f, W_g: Linear(2H -> H);
W_o: Linear(H -> 2); xavier-uniform weights, zero biases (pgjanet.py:79-84).  Kernels: csrc/janet_family.hip.
Deviation: the reference registry cannot build this class (models.py:109-114 passes window_size=, which
PGJANET.__init__ does not accept -> TypeError); here the registry name constructs it, as obviously intended."""
import torch.nn as nn

from .native import NativeBackbone


class PGJANET(NativeBackbone):
    backbone_name = "pgjanet"

    def __init__(self, hidden_size, output_size, bias=True):
        super().__init__()
        if not bias or output_size != 2:
            raise NotImplementedError("pgjanet kernels implement bias=True, I/Q output")
        self.hidden_size, self.output_size, self.bias, self.num_layers = hidden_size, output_size, bias, 1
        self.W_a = nn.Linear(hidden_size + 1, hidden_size, bias=True)
        self.W_p1 = nn.Linear(hidden_size + 1, hidden_size, bias=True)
        self.W_p2 = nn.Linear(hidden_size + 1, hidden_size, bias=True)
        self.W_f = nn.Linear(2 * hidden_size, hidden_size, bias=True)
        self.W_g = nn.Linear(2 * hidden_size, hidden_size, bias=True)
        self.W_o = nn.Linear(hidden_size, output_size, bias=True)
        self._finalize(hidden_size)

    def reset_parameters(self):
        for m in (self.W_a, self.W_p1, self.W_p2, self.W_f, self.W_g, self.W_o):
            nn.init.xavier_uniform_(m.weight)
            nn.init.constant_(m.bias, 0)
