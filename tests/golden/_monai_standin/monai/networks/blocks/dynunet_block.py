"""Stand-in for monai.networks.blocks.dynunet_block.UnetOutBlock (hybrid_CTUNet.py:19,781-783,810):
a 1x1x1 convolution with bias, registered as self.conv = Convolution(...) (keys '<name>.conv.conv.weight/bias')."""
import torch.nn as nn

from .convolutions import Convolution


class UnetOutBlock(nn.Module):
    def __init__(self, spatial_dims, in_channels, out_channels, dropout=None):
        super().__init__()
        self.conv = Convolution(spatial_dims, in_channels, out_channels, strides=1, kernel_size=1,
                                bias=True, conv_only=True, padding=0)

    def forward(self, inp):
        return self.conv(inp)
