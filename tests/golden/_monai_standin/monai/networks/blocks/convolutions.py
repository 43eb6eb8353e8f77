"""Stand-in for monai.networks.blocks.convolutions.Convolution as called by
networks/resnet.py:35-50 of the reference: always conv_only=True, dropout None."""
import torch.nn as nn


class Convolution(nn.Sequential):
    def __init__(self, dimensions, in_channels, out_channels, strides=1, kernel_size=3, act=None, norm=None,
                 dropout=None, dropout_dim=1, dilation=1, groups=1, bias=True, conv_only=False,
                 is_transposed=False, padding=None, output_padding=None):
        super().__init__()
        if dimensions != 3 or not conv_only or dropout is not None:
            raise NotImplementedError("stand-in covers only the 3-D conv_only=True, dropout=None use")
        if padding is None:
            raise NotImplementedError("stand-in expects an explicit padding (the reference always passes one)")
        if is_transposed:
            if output_padding is None:
                output_padding = 0
            conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size=kernel_size, stride=strides,
                                      padding=padding, output_padding=output_padding, groups=groups,
                                      bias=bias, dilation=dilation)
        else:
            conv = nn.Conv3d(in_channels, out_channels, kernel_size=kernel_size, stride=strides,
                             padding=padding, dilation=dilation, groups=groups, bias=bias)
        self.add_module("conv", conv)
