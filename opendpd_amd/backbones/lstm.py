"""HIP-backed LSTM-cell backbones: lstm, vdlstm.

Reference behaviour reproduced (parameter names / shapes / init / maths):
  LSTM    backbones/lstm.py:4-48     — rnn.{weight_ih_l0 (4H,2), weight_hh_l0 (4H,H), bias_*}, fc_out (2,H);
                                       the reference passes (h_0, h_0) as initial (h, c): both zero.
  VDLSTM  backbones/vdlstm.py:5-111  — rnn (input = window of 4 amplitudes), fc_lambda_1/2 (4,H), fc_out (2,8);
                                       the h_0 argument is ignored (vdlstm.py:77).
Kernels: opendpd_amd/csrc/lstm_family.hip.
"""
import torch.nn as nn

from .. import _lib
from .gru import _check_layers, _check_single_layer
from .native import NativeBackbone, RnnParams, init_gatewise, init_linear


class LSTM(NativeBackbone):
    backbone_name = "lstm"

    def __init__(self, input_size, hidden_size, output_size, num_layers, bidirectional=False, batch_first=True,
                 bias=True):
        super().__init__()
        _check_layers(num_layers, bidirectional, hidden_size)      # one layer, or two of <= 32 units (csrc/lstm_layers2.hip)
        if input_size != 2 or output_size != 2 or not bias:
            raise NotImplementedError("lstm backbone: input/output are I/Q pairs with bias (models.py:12-24)")
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, input_size, output_size, num_layers
        self.rnn = RnnParams(input_size, hidden_size, gates=4, num_layers=num_layers)
        self.fc_out = nn.Linear(hidden_size, output_size, bias=True)
        self._finalize(hidden_size)
        if num_layers == 2:
            self.desc.flags |= _lib.FLAG_TWO_LAYERS

    def reset_parameters(self):
        init_gatewise(self.rnn, self.hidden_size)
        init_linear(self.fc_out, "xavier")


class VDLSTM(NativeBackbone):
    backbone_name = "vdlstm"

    def __init__(self, input_size, hidden_size, output_size, num_layers, window_length=4, stride=1, bidirectional=False,
                 batch_first=True, bias=True):
        super().__init__()
        _check_single_layer(num_layers, bidirectional)
        if window_length != 4 or stride != 1:
            raise NotImplementedError("vdlstm kernels implement the reference default window_length=4, stride=1")
        self.hidden_size, self.input_size, self.output_size, self.num_layers = hidden_size, window_length, output_size, 1
        self.window_length, self.stride, self.pad_size = window_length, stride, window_length - 1
        self.rnn = RnnParams(window_length, hidden_size, gates=4)
        self.fc_lambda_1 = nn.Linear(hidden_size, window_length, bias=True)
        self.fc_lambda_2 = nn.Linear(hidden_size, window_length, bias=True)
        self.fc_out = nn.Linear(2 * window_length, 2, bias=True)
        self._finalize(hidden_size)

    def reset_parameters(self):
        init_gatewise(self.rnn, self.hidden_size)
        init_linear(self.fc_lambda_1, "xavier")
        init_linear(self.fc_lambda_2, "xavier")
        init_linear(self.fc_out, "xavier")
