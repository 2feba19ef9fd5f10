"""HIP-backed MCLDNN backbone (reference backbones/mcldnn.py:9-134).

Parameters (names / shapes / init as the reference): conv2d_1 = Conv2d(1 -> C, 3x3, pad 1), conv1d = Conv1d(5 -> 5C, k3, pad 1,
groups 5), conv2d_2 = Conv2d(10 -> 1, 3x3, pad 1), lstm = nn.LSTM(5C -> 8), fc_out = Linear(8 -> 16), fc_out_2 = Linear(16 -> 2);
every tensor whose name contains 'weight' xavier-uniform, every 'bias' zero (mcldnn.py:31-37; the constructor initialises once, the
registry's reset_parameters() a second time — both draws are kept so that a seeded construction consumes the generator as the
reference does).  The torch modules are parameter holders only; the arithmetic is csrc/mcldnn.hip (hidden = channels C <= 16)."""
import torch.nn as nn

from .native import NativeBackbone

MAX_HIDDEN = 16


class MCLDNN(NativeBackbone):
    backbone_name = "mcldnn"

    def __init__(self, hidden_size=8):
        super().__init__()
        if hidden_size > MAX_HIDDEN:
            raise NotImplementedError(f"mcldnn kernels cover hidden_size (channels) <= {MAX_HIDDEN}")
        C = self.channels = self.hidden_size = hidden_size
        self.memory_length, self.order, self.input_height, self.input_width, self.kernel_size = 5, 3, 5, 5, 3
        self.output_size, self.num_layers = 2, 1
        self.conv2d_1 = nn.Conv2d(1, C, kernel_size=3, padding=1)
        self.conv1d = nn.Conv1d(5, 5 * C, kernel_size=3, padding=1, groups=5)
        self.conv2d_2 = nn.Conv2d(10, 1, kernel_size=3, padding=1)
        self.lstm = nn.LSTM(input_size=5 * C, hidden_size=8, num_layers=1, batch_first=True)
        self.fc_out = nn.Linear(8, 16)
        self.fc_out_2 = nn.Linear(16, 2)
        self.reset_parameters()
        self._finalize(hidden_size)

    def reset_parameters(self):
        for name, p in self.named_parameters():
            if "weight" in name:
                nn.init.xavier_uniform_(p)
            elif "bias" in name:
                nn.init.constant_(p, 0)
