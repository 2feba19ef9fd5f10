"""HIP-backed GRU-cell backbones: gru, dgru, qgru, qgru_amp1.

Reference behaviour reproduced (parameter names / shapes / init / maths):
  GRU   backbones/gru.py:4-48   — rnn.{weight_ih_l0 (3H,2), weight_hh_l0 (3H,H), bias_*}, fc_out (2,H)
  DGRU  backbones/dgru.py:9-74  — rnn (input 6), fc_out (2,H+6), fc_hid (H,H)   [fc_out registered first]
  QGRU  backbones/qgru.py:9-71, qgru_amp1.py:9-76 — rnn (input 4), fc_out (2,H)
The time recurrence, feature extraction and output heads run in opendpd_amd/csrc/gru_family.hip.
"""
import torch.nn as nn

from .. import _lib

from .native import NativeBackbone, RnnParams, init_gatewise, init_linear


def _check_layers(num_layers, bidirectional, hidden_size):
    """one layer, or two stacked layers of <= 32 units (csrc/gru_layers2.hip: both layers in one wave, time-skewed)"""
    if bidirectional or num_layers not in (1, 2) or (num_layers == 2 and hidden_size > 32):
        raise NotImplementedError("the HIP GRU kernels implement one layer, or two layers of <= 32 units, unidirectional")


def _check_single_layer(num_layers, bidirectional):
    if num_layers != 1 or bidirectional:
        raise NotImplementedError("the HIP recurrent kernels implement num_layers=1, unidirectional "
                                  "(the only configuration the reference scripts use)")


class GRU(NativeBackbone):
    backbone_name = "gru"

    def __init__(self, input_size, hidden_size, output_size, num_layers, bidirectional=False, batch_first=True,
                 bias=True):
        super().__init__()
        _check_layers(num_layers, bidirectional, hidden_size)
        if input_size != 2 or output_size != 2 or not bias:
            raise NotImplementedError("gru backbone: input/output are I/Q pairs with bias (models.py:12-24)")
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, input_size, output_size, num_layers
        self.rnn = RnnParams(input_size, hidden_size, gates=3, num_layers=num_layers)
        self.fc_out = nn.Linear(hidden_size, output_size, bias=True)
        self._finalize(hidden_size)
        if num_layers == 2:
            self.desc.flags |= _lib.FLAG_TWO_LAYERS

    def reset_parameters(self):
        init_gatewise(self.rnn, self.hidden_size)
        init_linear(self.fc_out, "xavier")


class DGRU(NativeBackbone):
    backbone_name = "dgru"

    def __init__(self, hidden_size, output_size, num_layers, bidirectional=False, batch_first=True, bias=True):
        super().__init__()
        _check_layers(num_layers, bidirectional, hidden_size)
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, 6, output_size, num_layers
        self.rnn = RnnParams(6, hidden_size, gates=3, num_layers=num_layers)
        self.fc_out = nn.Linear(hidden_size + 6, output_size, bias=True)
        self.fc_hid = nn.Linear(hidden_size, hidden_size, bias=True)
        self._finalize(hidden_size)
        if num_layers == 2:
            self.desc.flags |= _lib.FLAG_TWO_LAYERS

    def reset_parameters(self):
        init_gatewise(self.rnn, self.hidden_size)
        init_linear(self.fc_out, "xavier")
        init_linear(self.fc_hid, "kaiming")


class QGRU(NativeBackbone):
    """Float path of backbones/qgru.py (features I,Q,|x|^2,|x|^4)."""
    backbone_name = "qgru"

    def __init__(self, hidden_size, output_size, num_layers, bidirectional=False, batch_first=True, bias=True):
        super().__init__()
        _check_layers(num_layers, bidirectional, hidden_size)
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, 4, output_size, num_layers
        self.rnn = RnnParams(4, hidden_size, gates=3, num_layers=num_layers)
        self.fc_out = nn.Linear(hidden_size, output_size, bias=True)
        self._finalize(hidden_size)
        if num_layers == 2:
            self.desc.flags |= _lib.FLAG_TWO_LAYERS

    def reset_parameters(self):
        # qgru.py:37-57: rnn + fc_out are re-initialised, then the reference touches a non-existent
        # self.fc_hid -> AttributeError, swallowed by CoreModel (models.py:144-148).  Net effect:
        init_gatewise(self.rnn, self.hidden_size)
        init_linear(self.fc_out, "xavier")


class QGRUAmp1(QGRU):
    """Float path of backbones/qgru_amp1.py (features I,Q,|x|,|x|^3)."""
    backbone_name = "qgru_amp1"
