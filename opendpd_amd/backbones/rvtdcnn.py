"""HIP-backed RVTDCNN backbone (reference backbones/rvtdcnn.py:9-62; the registry passes fc_hid_size = hidden_size, models.py:80-81).

Same modules, names and construction order as the reference (`Conv2d`, `fc_hid`, `fc_out` with PyTorch's default initialisation —
the reference defines no reset_parameters for this class), so a seeded construction gives the reference's state dict.
Kernels: csrc/rvtdcnn.hip (window 4, 3 channels, 3 x 3 kernel — the only configuration the registry builds — fc_hid_size <= 32;
larger sizes run as the ATen restatement of backbones/extras.py, said aloud like every configuration outside the envelope)."""
import torch.nn as nn

from .native import NativeBackbone


class RVTDCNN(NativeBackbone):
    backbone_name = "rvtdcnn"

    def __init__(self, window_size=4, out_channels=3, kernel_size=3, stride=1, padding=(1, 0), dilation=1, fc_hid_size=6):
        super().__init__()
        if (window_size, out_channels, kernel_size, stride, tuple(padding), dilation) != (4, 3, 3, 1, (1, 0), 1):
            raise NotImplementedError("the HIP kernels cover the configuration the registry builds (window 4, 3 channels, k3, padding (1,0))")
        self.window_size, self.out_channels, self.fc_hid_size = window_size, out_channels, fc_hid_size
        self.stride, self.feature_size_new = stride, 3
        self.fc_in_features = out_channels * 3 * window_size
        self.Conv2d = nn.Conv2d(1, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation, bias=True)
        self.fc_hid = nn.Linear(self.fc_in_features, fc_hid_size)
        self.fc_out = nn.Linear(fc_hid_size, 2)
        self._finalize(fc_hid_size)
