"""3-D ResNet encoder on MI355X kernels — host-side mirror of the reference's networks/resnet.py.

Same constructor arguments, module tree and state_dict keys as the reference (conv weights live in a child named
``conv``, exactly as MONAI's ``Convolution(conv_only=True)`` registers them; InstanceNorm has no parameters).
torch.nn.Conv3d / ConvTranspose3d modules are used only as PARAMETER HOLDERS (identical default init and keys);
their forward is never called: all arithmetic goes through ..ops into libctunet_hip.so, channels-last.
"""
from __future__ import annotations

from typing import Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from .. import ops, ops_fused


def get_inplanes():
    return [32, 64, 128, 256]  # networks/resnet.py:14-15


def _t3(v) -> Tuple[int, int, int]:
    return tuple(int(x) for x in v) if isinstance(v, (tuple, list)) else (int(v),) * 3


def get_padding(kernel_size, stride):
    """networks/resnet.py:52-64 (same AssertionError on negative padding)."""
    kernel_size_np = np.atleast_1d(kernel_size)
    stride_np = np.atleast_1d(stride)
    padding_np = (kernel_size_np - stride_np + 1) / 2
    if np.min(padding_np) < 0:
        raise AssertionError("padding value should not be negative, please change the kernel size and/or stride.")
    padding = tuple(int(p) for p in padding_np)
    return padding if len(padding) > 1 else padding[0]


def get_output_padding(kernel_size, stride, padding):
    """networks/resnet.py:66-80."""
    kernel_size_np = np.atleast_1d(kernel_size)
    stride_np = np.atleast_1d(stride)
    padding_np = np.atleast_1d(padding)
    out_padding_np = 2 * padding_np + stride_np - kernel_size_np
    if np.min(out_padding_np) < 0:
        raise AssertionError("out_padding value should not be negative, please change the kernel size and/or stride.")
    out_padding = tuple(int(p) for p in out_padding_np)
    return out_padding if len(out_padding) > 1 else out_padding[0]


class ConvLayer(nn.Module):
    """What get_conv_layer(..., conv_only=True) returns (networks/resnet.py:17-50): a container whose single child
    ``conv`` owns the weight.  forward(x, x2=None) takes channels-last volumes; x2 is an optional second tensor that
    the reference would torch.cat along channels before the convolution."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, kernel_size=3, stride=1,
                 bias: bool = False, is_transposed: bool = False):
        super().__init__()
        if spatial_dims != 3:
            raise NotImplementedError("hybrid-ctunet_amd implements the 3-D path only")
        k, s = _t3(kernel_size), _t3(stride)
        p = _t3(get_padding(k, s))
        self.kernel_size, self.stride, self.padding = k, s, p
        self.in_channels, self.out_channels, self.is_transposed = in_channels, out_channels, is_transposed
        if is_transposed:
            op = _t3(get_output_padding(k, s, p))
            if k != s or any(p) or any(op):
                raise NotImplementedError("transposed convolution is implemented for kernel == stride, padding 0 "
                                          "(the only form the reference uses)")
            self.conv = nn.ConvTranspose3d(in_channels, out_channels, k, stride=s, padding=p, output_padding=op,
                                           bias=bias)
        else:
            self.conv = nn.Conv3d(in_channels, out_channels, k, stride=s, padding=p, bias=bias)

    def forward(self, x, x2=None, grad_stash=None, grad_stash2=None):
        w = self.conv.weight
        if self.is_transposed:
            assert x2 is None
            return ops.conv_transpose3d(x, w, grad_stash=grad_stash)
        if self.in_channels == 1:
            assert x2 is None
            return ops.conv3d_cin1(x, w, self.stride, self.padding)
        if x2 is None and self.kernel_size == (1, 1, 1) and self.stride == (1, 1, 1):
            # every 1x1x1 ConvLayer of these networks feeds an InstanceNorm
            return ops.linear(x, w, in_stats=True, grad_stash=grad_stash)
        return ops.conv3d(x, w, self.stride, self.padding, x2, grad_stash=grad_stash, grad_stash2=grad_stash2)


def get_conv_layer(spatial_dims: int, in_channels: int, out_channels: int, kernel_size=3, stride=1, act=None,
                   norm=None, dropout=None, groups: int = 1, bias: bool = False, conv_only: bool = True,
                   is_transposed: bool = False):
    """Same signature as networks/resnet.py:17-30; only the combination the reference uses is implemented."""
    if not conv_only or dropout is not None or groups != 1:
        raise NotImplementedError("only conv_only=True, dropout=None, groups=1 (as used by the reference)")
    return ConvLayer(spatial_dims, in_channels, out_channels, kernel_size, stride, bias, is_transposed)


def check_norm(norm_name):
    name = norm_name[0] if isinstance(norm_name, (tuple, list)) else norm_name
    if str(name).lower() != "instance":
        raise NotImplementedError(f"norm_name={norm_name!r}: only 'instance' (the reference default, non-affine "
                                  "InstanceNorm3d) has an MI355X kernel")


class _Downsample(nn.Sequential):
    """networks/resnet.py:196-199: Sequential(conv1x1x1(stride), InstanceNorm3d) -> key 'downsample.0.conv.weight'."""

    def __init__(self, cin, cout, stride):
        super().__init__(ConvLayer(3, cin, cout, 1, stride), nn.Identity())

    def forward(self, x):
        return ops.instance_norm(self[0](x), None, False)


class Bottleneck(nn.Module):
    """networks/resnet.py:82-126."""
    expansion = 4

    def __init__(self, in_planes: int, planes: int, spatial_dims: int = 3, stride=1, norm_name="instance",
                 dropout=None, downsample=None):
        super().__init__()
        check_norm(norm_name)
        self.conv1 = get_conv_layer(spatial_dims, in_planes, planes, kernel_size=1, stride=1, dropout=dropout)
        self.conv2 = get_conv_layer(spatial_dims, planes, planes, kernel_size=3, stride=stride, dropout=dropout)
        self.conv3 = get_conv_layer(spatial_dims, planes, planes * self.expansion, kernel_size=1, stride=1,
                                    dropout=dropout)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if ops_fused.bottleneck_ok(self, x):
            return ops_fused.bottleneck(self, x)   # one autograd node, forward / backward replayed from launch lists
        # shortcut (identity, or conv + norm in the first block of a stage): its gradient w.r.t. x is parked (GradStash) and
        # added inside conv1's data-gradient GEMM instead of by an autograd accumulation pass
        stash = [] if (x.requires_grad and torch.is_grad_enabled()) else None
        y1 = self.conv1(x, grad_stash=stash)
        c2 = self.conv2
        # gn1's output feeds conv2 only: written 16-channel-blocked when conv2 runs on the halo kernel (ops.wants_b16)
        out = ops.instance_norm(y1, None, True, out_b16=ops.wants_b16(c2.conv.weight, y1, c2.stride, c2.padding))
        out = ops.instance_norm(self.conv2(out), None, True)
        out = self.conv3(out)
        if self.downsample is not None:
            residual = self.downsample(ops.GradStash.apply(x, stash) if stash is not None else x)
        else:
            # identity shortcut: its gradient is folded into conv1's data-gradient GEMM (ops.GradStash)
            residual = ops.GradStash.apply(x, stash) if stash is not None else x
        return ops.instance_norm(out, residual, True)  # gn3 -> += residual -> LeakyReLU (resnet.py:118-124)


class ResNet(nn.Module):
    """networks/resnet.py:128-230.  forward(x) takes a channels-last [B, D, H, W, 1] volume and returns the four stage
    outputs channels-last."""

    def __init__(self, block, layers: Sequence[int], block_inplanes: Sequence[int], shortcut_type: str = "B",
                 n_input_channels: int = 1, conv1_t_size: int = 7, DS_stride=((2, 2, 1), (2, 2, 2), (2, 2, 2), (2, 2, 2)),
                 no_max_pool: bool = True, width_factor: float = 1.0, spatial_dims: int = 3, norm_name="instance"):
        super().__init__()
        check_norm(norm_name)
        if shortcut_type != "B" or not no_max_pool or n_input_channels != 1:
            raise NotImplementedError("only shortcut 'B', no_max_pool=True, one input channel (the reference's use)")
        block_inplanes = [int(x * width_factor) for x in block_inplanes]
        self.in_planes = 64
        self.no_max_pool = no_max_pool
        self.conv1 = get_conv_layer(spatial_dims, n_input_channels, self.in_planes, kernel_size=(7, 7, conv1_t_size),
                                    stride=DS_stride[0])
        self.layer1 = self._make_layer(block, block_inplanes[0], layers[0], shortcut_type)
        self.layer2 = self._make_layer(block, block_inplanes[1], layers[1], shortcut_type, stride=DS_stride[1])
        self.layer3 = self._make_layer(block, block_inplanes[2], layers[2], shortcut_type, stride=DS_stride[2])
        self.layer4 = self._make_layer(block, block_inplanes[3], layers[3], shortcut_type, stride=DS_stride[3])

    def _make_layer(self, block, planes, blocks, shortcut_type, stride=1):
        downsample = None
        if stride != 1 or self.in_planes != planes * block.expansion:
            downsample = _Downsample(self.in_planes, planes * block.expansion, stride)
        layers = [block(in_planes=self.in_planes, planes=planes, stride=stride, downsample=downsample)]
        self.in_planes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.in_planes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = ops.trace_point(ops.instance_norm(self.conv1(x), None, True), "stem")
        features = []
        for i, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4)):
            x = ops.trace_point(layer(x), f"layer{i + 1}")
            features.append(x)
        return features


def generate_model(model_depth, **kwargs):
    """networks/resnet.py:233-245 (note the non-standard 101 layout [8, 9, 13, 3])."""
    assert model_depth in [50, 101, 152, 200]
    layers = {50: [3, 4, 6, 3], 101: [8, 9, 13, 3], 152: [8, 9, 30, 3], 200: [8, 25, 30, 3]}[model_depth]
    return ResNet(Bottleneck, layers, get_inplanes(), **kwargs)
