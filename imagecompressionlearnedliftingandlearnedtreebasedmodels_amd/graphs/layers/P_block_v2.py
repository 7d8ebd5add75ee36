"""P_block_v2 -- the predict/update CNN of the learned lifting (reference graphs/layers/P_block_v2.py:7-55).

Parameter container with the reference's names (conv1..conv4) and default initialisation; the arithmetic runs fused
inside the lifting-step kernels (csrc/lifting.hip).  ``forward`` evaluates the block alone through the conv kernels.
"""
import torch.nn as nn

from ... import ops


class P_block_v2(nn.Module):
    def __init__(self, linearity_flag=1, csize=1, conv_filter_size=3, depth_scale=16):
        super().__init__()
        k = self.conv_filter_size = conv_filter_size
        self.padding = k // 2
        self.csize = csize
        d = depth_scale * csize
        self.conv1 = nn.Conv2d(csize, d, k, stride=1, padding=self.padding)
        self.conv2 = nn.Conv2d(d, d, k, stride=1, padding=self.padding)
        self.conv3 = nn.Conv2d(d, d, k, stride=1, padding=self.padding)
        self.conv4 = nn.Conv2d(d, csize, k, stride=1, padding=self.padding)
        self.linearityFlag = linearity_flag

    def forward(self, tmp):
        """conv1 -> tanh -> conv2 -> tanh -> conv3 (+ conv1 pre-activation) -> conv4 (P_block_v2.py:40-55)."""
        act = ops.ACT_TANH if self.linearityFlag == 1 else ops.ACT_NONE
        k = self.conv_filter_size
        x = tmp[None].contiguous()
        c = lambda m: (m.weight.detach()[None].contiguous(), m.bias.detach()[None].contiguous())
        r = ops.conv2d(x, *c(self.conv1), k)
        t = ops.conv2d(x, *c(self.conv1), k, act=act)
        t = ops.conv2d(t, *c(self.conv2), k, act=act)
        t = ops.conv2d(t, *c(self.conv3), k) + r
        return ops.conv2d(t, *c(self.conv4), k)[0]
