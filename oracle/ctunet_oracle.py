"""ORACLE (test infrastructure, NOT product code): CPU restatement of Hybrid-CTUNet's volumetric hot path.

Plain PyTorch (fp32/fp64, CPU), no MONAI, no einops.  Every class cites the reference file:line it follows
(paths relative to the reference checkout).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product package ``hybrid-ctunet_amd`` never does.

Pinning: this restatement is checked against the reference's own ``networks/*.py`` imported unchanged in the
build container (``tests/golden/make_golden.py``), and against the golden vectors that script commits under
``tests/golden/``.  The reference has no tests/fixtures of its own.  The six MONAI 0.7.0
wrapper symbols the reference imports are not installed; their construction semantics are restated
(``tests/golden/_monai_standin``), so parity is **unpinned at the MONAI boundary** (InstanceNorm3d non-affine;
child module name ``conv``), pinned everywhere else by executing the reference.

state_dict keys/shapes are identical to the reference modules (412 / 126 / 235 tensors for CTUNet d101 pf8 /
CUNet d101 / TUNet pf8), which is what makes weights interchangeable between oracle, reference and product.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

LRELU_SLOPE = 0.01  # networks/resnet.py:102, networks/hybrid_CTUNet.py:84
IN_EPS = 1e-5       # torch.nn.InstanceNorm3d default (see _monai_standin caveat)


# ----------------------------------------------------------------------------------------------------------
# conv factory  (networks/resnet.py:17-80)
# ----------------------------------------------------------------------------------------------------------
def _t3(v) -> Tuple[int, int, int]:
    return tuple(int(x) for x in v) if isinstance(v, (tuple, list)) else (int(v),) * 3


def get_padding(kernel_size, stride) -> Tuple[int, int, int]:
    """networks/resnet.py:52-64: padding = int((k - s + 1) / 2) per dim; negative -> AssertionError."""
    k, s = _t3(kernel_size), _t3(stride)
    p = [(kk - ss + 1) / 2 for kk, ss in zip(k, s)]
    if min(p) < 0:
        raise AssertionError("padding value should not be negative, please change the kernel size and/or stride.")
    return tuple(int(x) for x in p)


def get_output_padding(kernel_size, stride, padding) -> Tuple[int, int, int]:
    """networks/resnet.py:66-80: output_padding = 2p + s - k per dim."""
    k, s, p = _t3(kernel_size), _t3(stride), _t3(padding)
    o = [2 * pp + ss - kk for kk, ss, pp in zip(k, s, p)]
    if min(o) < 0:
        raise AssertionError("out_padding value should not be negative, please change the kernel size and/or stride.")
    return tuple(int(x) for x in o)


class ConvLayer(nn.Module):
    """get_conv_layer(..., conv_only=True) (networks/resnet.py:17-50): one Conv3d/ConvTranspose3d child 'conv'."""

    def __init__(self, cin, cout, kernel_size=3, stride=1, bias=False, is_transposed=False):
        super().__init__()
        k, s = _t3(kernel_size), _t3(stride)
        p = get_padding(k, s)
        if is_transposed:
            op = get_output_padding(k, s, p)
            self.conv = nn.ConvTranspose3d(cin, cout, k, stride=s, padding=p, output_padding=op, bias=bias)
        else:
            self.conv = nn.Conv3d(cin, cout, k, stride=s, padding=p, bias=bias)

    def forward(self, x):
        return self.conv(x)


def inorm(x):
    """InstanceNorm3d, eps 1e-5, no affine, no running stats."""
    return F.instance_norm(x, eps=IN_EPS)


def lrelu(x):
    return F.leaky_relu(x, LRELU_SLOPE)


# ----------------------------------------------------------------------------------------------------------
# 3-D ResNet encoder  (networks/resnet.py:82-245)
# ----------------------------------------------------------------------------------------------------------
class _Downsample(nn.Sequential):
    """networks/resnet.py:196-199: Sequential(conv1x1x1(stride), InstanceNorm) -> key 'downsample.0.conv.weight'."""

    def __init__(self, cin, cout, stride):
        super().__init__(ConvLayer(cin, cout, 1, stride), nn.Identity())

    def forward(self, x):
        return inorm(self[0](x))


class Bottleneck(nn.Module):
    """networks/resnet.py:82-126."""
    expansion = 4

    def __init__(self, in_planes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = ConvLayer(in_planes, planes, 1, 1)
        self.conv2 = ConvLayer(planes, planes, 3, stride)
        self.conv3 = ConvLayer(planes, planes * 4, 1, 1)
        self.downsample = downsample

    def forward(self, x):
        out = lrelu(inorm(self.conv1(x)))
        out = lrelu(inorm(self.conv2(out)))
        out = inorm(self.conv3(out))
        residual = x if self.downsample is None else self.downsample(x)
        return lrelu(out + residual)


RESNET_LAYERS = {50: [3, 4, 6, 3], 101: [8, 9, 13, 3], 152: [8, 9, 30, 3], 200: [8, 25, 30, 3]}  # resnet.py:236-243
DS_STRIDE = ((2, 2, 1), (2, 2, 2), (2, 2, 2), (2, 2, 2))  # hybrid_CTUNet.py:728


class ResNet(nn.Module):
    """networks/resnet.py:128-230 (no_max_pool=True, shortcut 'B', widths 32/64/128/256 x4)."""

    def __init__(self, model_depth: int, DS_stride=DS_STRIDE):
        super().__init__()
        assert model_depth in [50, 101, 152, 200]  # resnet.py:234
        layers = RESNET_LAYERS[model_depth]
        self.in_planes = 64
        self.conv1 = ConvLayer(1, 64, (7, 7, 7), DS_stride[0])
        self.layer1 = self._make_layer(32, layers[0], 1)
        self.layer2 = self._make_layer(64, layers[1], DS_stride[1])
        self.layer3 = self._make_layer(128, layers[2], DS_stride[2])
        self.layer4 = self._make_layer(256, layers[3], DS_stride[3])

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.in_planes != planes * 4:
            downsample = _Downsample(self.in_planes, planes * 4, stride)
        mods = [Bottleneck(self.in_planes, planes, stride, downsample)]
        self.in_planes = planes * 4
        for _ in range(1, blocks):
            mods.append(Bottleneck(self.in_planes, planes))
        return nn.Sequential(*mods)

    def forward(self, x):
        x = lrelu(inorm(self.conv1(x)))
        feats = []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = layer(x)
            feats.append(x)
        return feats


# ----------------------------------------------------------------------------------------------------------
# conv blocks  (networks/hybrid_CTUNet.py:29-255, 593-620)
# ----------------------------------------------------------------------------------------------------------
class ResBlock(nn.Module):
    """networks/hybrid_CTUNet.py:29-105.  conv3 is always constructed, used only when in!=out or stride!=1."""

    def __init__(self, cin, cout, kernel_size=3, stride=1):
        super().__init__()
        self.conv1 = ConvLayer(cin, cout, kernel_size, stride)
        self.conv2 = ConvLayer(cout, cout, kernel_size, 1)
        self.conv3 = ConvLayer(cin, cout, 1, stride)
        self.downsample = (cin != cout) or any(s != 1 for s in _t3(stride))

    def forward(self, inp):
        out = lrelu(inorm(self.conv1(inp)))
        out = inorm(self.conv2(out))
        residual = inorm(self.conv3(inp)) if self.downsample else inp
        return lrelu(out + residual)


class BasicConvBlock(nn.Module):
    """networks/hybrid_CTUNet.py:107-146."""

    def __init__(self, cin, cout, kernel_size, stride):
        super().__init__()
        self.layer = ResBlock(cin, cout, kernel_size, stride)

    def forward(self, x):
        return self.layer(x)


class UpCatConvBlock(nn.Module):
    """networks/hybrid_CTUNet.py:148-201."""

    def __init__(self, cin, cout, kernel_size, upsample_kernel_size):
        super().__init__()
        self.transp_conv = ConvLayer(cin, cout, upsample_kernel_size, upsample_kernel_size, is_transposed=True)
        self.conv_block = ResBlock(cout + cout, cout, kernel_size, 1)

    def forward(self, inp, skip):
        return self.conv_block(torch.cat((self.transp_conv(inp), skip), dim=1))


class UpConvBlock(nn.Module):
    """networks/hybrid_CTUNet.py:203-255."""

    def __init__(self, cin, cout, kernel_size, upsample_kernel_size):
        super().__init__()
        self.transp_conv = ConvLayer(cin, cout, upsample_kernel_size, upsample_kernel_size, is_transposed=True)
        self.conv_block = ResBlock(cout, cout, kernel_size, 1)

    def forward(self, inp):
        return self.conv_block(self.transp_conv(inp))


class CatConvBlock(nn.Module):
    """networks/hybrid_CTUNet.py:593-620."""

    def __init__(self, cin, kernel_size):
        super().__init__()
        self.conv_block = ResBlock(cin + cin, cin, kernel_size, 1)

    def forward(self, x, skip):
        return self.conv_block(torch.cat((x, skip), dim=1))


# ----------------------------------------------------------------------------------------------------------
# binary cross-weight fusion  (networks/hybrid_CTUNet.py:622-669)
# ----------------------------------------------------------------------------------------------------------
class PixelweightAttention(nn.Module):
    def __init__(self, dim, dim_head=32):
        super().__init__()
        self.dim_head = dim_head
        self.heads = dim // dim_head
        self.scale = dim_head ** -0.5
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.to_qkv1 = nn.Linear(dim, dim * 3, bias=False)
        self.to_qkv2 = nn.Linear(dim, dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(dim, dim, bias=False), nn.Identity())

    def forward(self, x1, x2):
        b, c, d0, d1, d2 = x1.shape
        n = d0 * d1 * d2
        t1 = self.norm1(x1.reshape(b, c, n).transpose(1, 2))          # 'b c f h w -> b (f h w) c' :648
        t2 = self.norm2(x2.reshape(b, c, n).transpose(1, 2))
        q1, k1, v1 = (t.reshape(b, n, self.heads, self.dim_head) for t in self.to_qkv1(t1).chunk(3, dim=-1))
        q2, k2, v2 = (t.reshape(b, n, self.heads, self.dim_head) for t in self.to_qkv2(t2).chunk(3, dim=-1))
        dots1 = (q2 * k1).sum(-1, keepdim=True) * self.scale            # :658
        dots2 = (q1 * k2).sum(-1, keepdim=True) * self.scale            # :659
        attn = torch.softmax(torch.cat((dots1, dots2), dim=-1), dim=-1)  # :660-661
        out = attn[..., 0:1] * v1 + attn[..., 1:2] * v2                  # :662-665
        out = self.to_out(out.reshape(b, n, c))
        return out.transpose(1, 2).reshape(b, c, d0, d1, d2)


class Up2FusionBlock(nn.Module):
    """networks/hybrid_CTUNet.py:257-341 (forward :329-341, 'fusion2'; forward_ is dead code)."""

    def __init__(self, cin, cout, kernel_size, upsample_kernel_size):
        super().__init__()
        self.transp_conv = ConvLayer(cin, cout, upsample_kernel_size, upsample_kernel_size, is_transposed=True)
        self.pixelweight_attention1 = PixelweightAttention(cout)
        self.pixelweight_attention2 = PixelweightAttention(cout)
        self.up_addconv_block1 = ResBlock(cout, cout, kernel_size, 1)
        self.up_addconv_block2 = ResBlock(cout, cout, kernel_size, 1)

    def forward(self, inp, skip_conv, skip_vit):
        skip = self.up_addconv_block1(self.pixelweight_attention1(skip_conv, skip_vit))
        out = self.transp_conv(inp)
        return self.up_addconv_block2(self.pixelweight_attention2(out, skip))


# ----------------------------------------------------------------------------------------------------------
# window attention / feed-forward / pixel shuffle  (networks/hybrid_CTUNet.py:388-591)
# ----------------------------------------------------------------------------------------------------------
def rel_pos_indices(window_size: int) -> torch.Tensor:
    """networks/hybrid_CTUNet.py:472-477: idx(i,j) = (hi-hj+w-1)*(2w-1)^2 + (wi-wj+w-1)*(2w-1) + (fi-fj+w-1)."""
    pos = torch.arange(window_size)
    grid = torch.stack(torch.meshgrid(pos, pos, pos, indexing="ij")).reshape(3, -1).t()  # (n, 3), order (h w f)
    rel = grid[:, None, :] - grid[None, :, :] + (window_size - 1)
    m = 2 * window_size - 1
    return (rel * torch.tensor([m * m, m, 1])).sum(-1)


class MultiAxisAttention(nn.Module):
    """networks/hybrid_CTUNet.py:442-511."""

    def __init__(self, dim, dim_head=32, window_size=6):
        super().__init__()
        assert dim % dim_head == 0, "dimension must be divisible by the head dimension"
        self.heads = dim // dim_head
        self.scale = dim_head ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.to_qkv = nn.Linear(dim, dim * 3, bias=False)
        self.attend = nn.Sequential(nn.Softmax(dim=-1), nn.Identity())  # Identity = nn.Dropout(0) (:459-462)
        self.to_out = nn.Sequential(nn.Linear(dim, dim, bias=False), nn.Identity())
        self.rel_pos_bias = nn.Embedding((2 * window_size - 1) ** 3, self.heads)
        self.register_buffer("rel_pos_indices", rel_pos_indices(window_size), persistent=False)

    def forward(self, x):  # x: (b, X, Y, Z, w1, w2, w3, d)
        b, X, Y, Z, w1, w2, w3, d = x.shape
        h = self.heads
        x = self.norm(x).reshape(b * X * Y * Z, w1 * w2 * w3, d)
        q, k, v = (t.reshape(t.shape[0], t.shape[1], h, d // h).transpose(1, 2) for t in self.to_qkv(x).chunk(3, -1))
        q = q * self.scale
        sim = q @ k.transpose(-1, -2)
        bias = self.rel_pos_bias(self.rel_pos_indices)  # (n, n, h)
        sim = sim + bias.permute(2, 0, 1)
        out = self.attend(sim) @ v                      # (B', h, n, dh)
        out = out.transpose(1, 2).reshape(b * X * Y * Z, w1, w2, w3, d)
        out = self.to_out(out)
        return out.reshape(b, X, Y, Z, w1, w2, w3, d)


class FeedForward(nn.Module):
    """networks/hybrid_CTUNet.py:513-526 and networks/vit.py:31-44 (same layout, exact-erf GELU)."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden), nn.GELU(), nn.Identity(),
                                 nn.Linear(hidden, dim), nn.Identity())

    def forward(self, x):
        return self.net(x)


class Residual(nn.Module):
    """networks/hybrid_CTUNet.py:434-440."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        return self.fn(x) + x


class PixelShuffle(nn.Module):
    """networks/hybrid_CTUNet.py:388-432 == 'b (c p1 p2 p3) h w f -> b (h p1) (w p2) (f p3) c' -> Linear -> chan-first."""

    def __init__(self, scale_factor, cin, cout):
        super().__init__()
        self.scale_factor = tuple(scale_factor)
        self.to_out = nn.Linear(cin // (scale_factor[0] * scale_factor[1] * scale_factor[2]), cout)

    def forward(self, x):
        b, c, d0, d1, d2 = x.shape
        p1, p2, p3 = self.scale_factor
        div = p1 * p2 * p3
        if c % div != 0:
            raise ValueError(f"Number of input channels ({c}) must be evenly divisible by {self.scale_factor}")
        oc = c // div
        x = x.reshape(b, oc, p1, p2, p3, d0, d1, d2).permute(0, 5, 2, 6, 3, 7, 4, 1)
        x = x.reshape(b, d0 * p1, d1 * p2, d2 * p3, oc)
        x = self.to_out(x)
        return x.permute(0, 4, 1, 2, 3)


class _Placeholder(nn.Identity):
    """Stands where the reference has an einops Rearrange layer (keeps nn.Sequential indices, no params)."""


def _partition(x, w, mode):
    """'b c (h h1)(w w1)(f f1) -> b h w f h1 w1 f1 c' (block, :559) or 'b c (h1 h)(w1 w)(f1 f) -> ...' (grid, :564)."""
    b, c, D0, D1, D2 = x.shape
    n0, n1, n2 = D0 // w, D1 // w, D2 // w
    if mode == "block":
        x = x.reshape(b, c, n0, w, n1, w, n2, w).permute(0, 2, 4, 6, 3, 5, 7, 1)
    else:
        x = x.reshape(b, c, w, n0, w, n1, w, n2).permute(0, 3, 5, 7, 2, 4, 6, 1)
    return x


def _unpartition(x, mode):
    b, n0, n1, n2, w, _, _, c = x.shape
    if mode == "block":
        x = x.permute(0, 7, 1, 4, 2, 5, 3, 6)
    else:
        x = x.permute(0, 7, 4, 1, 5, 2, 6, 3)
    return x.reshape(b, c, n0 * w, n1 * w, n2 * w)


class UpAttentionBlock(nn.Module):
    """networks/hybrid_CTUNet.py:528-591 (depth (1,1,1,1), window 6, dropout 0)."""

    def __init__(self, in_channels, dims=(128, 256, 512, 1024), DS_stride=DS_STRIDE):
        super().__init__()
        dims = (in_channels, *dims[::-1][1:], 64)  # :546 -> (768, 512, 256, 128, 64)
        self.layers = nn.ModuleList()
        w = 6
        for ind, (din, dout) in enumerate(zip(dims[:-1], dims[1:])):
            if ind <= 2:
                block = nn.Sequential(
                    _Placeholder(),
                    Residual(MultiAxisAttention(din, 32, w)), Residual(FeedForward(din, din * 4)),
                    _Placeholder(), _Placeholder(),
                    Residual(MultiAxisAttention(din, 32, w)), Residual(FeedForward(din, din * 4)),
                    _Placeholder(),
                    PixelShuffle(DS_stride[::-1][ind], din, dout),
                )
            else:
                block = nn.Sequential(
                    _Placeholder(), Residual(FeedForward(din, din * 4)), Residual(FeedForward(din, din * 4)),
                    _Placeholder(), PixelShuffle(DS_stride[::-1][ind], din, dout),
                )
            self.layers.append(nn.Sequential(block))
        self.window = w

    def _stage(self, ind, x):
        blk = self.layers[ind][0]
        if ind <= 2:
            x = _partition(x, self.window, "block")
            x = blk[1](x)
            x = blk[2](x)
            x = _unpartition(x, "block")
            x = _partition(x, self.window, "grid")
            x = blk[5](x)
            x = blk[6](x)
            x = _unpartition(x, "grid")
            return blk[8](x)
        x = x.permute(0, 2, 3, 4, 1)
        x = blk[1](x)
        x = blk[2](x)
        x = x.permute(0, 4, 1, 2, 3)
        return blk[4](x)

    def forward(self, x):
        feats = [x]
        for ind in range(len(self.layers)):
            x = self._stage(ind, x)
            feats.append(x)
        return feats


class DecoderLinear(nn.Module):
    """networks/hybrid_CTUNet.py:671-691 (patch_size 1)."""

    def __init__(self, n_cls, d_encoder):
        super().__init__()
        self.head = nn.Linear(d_encoder, n_cls)

    def forward(self, x, im_size):
        b = x.shape[0]
        x = self.head(x)
        return x.transpose(1, 2).reshape(b, -1, *im_size)


class UnetOutBlock(nn.Module):
    """MONAI UnetOutBlock as used at hybrid_CTUNet.py:781-783,810: 1x1x1 conv with bias, keys '<n>.conv.conv.*'."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = ConvLayer(cin, cout, 1, 1, bias=True)

    def forward(self, x):
        return self.conv(x)


# ----------------------------------------------------------------------------------------------------------
# 3-D ViT  (networks/vit.py:31-139)
# ----------------------------------------------------------------------------------------------------------
class Attention(nn.Module):
    """networks/vit.py:46-78 (pre-LN inside, qkv no bias, out-proj with bias, scale on q.k^T)."""

    def __init__(self, dim, heads=8, dim_head=64):
        super().__init__()
        inner = dim_head * heads
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.dropout = nn.Identity()  # nn.Dropout(0) on the attention probabilities (vit.py:57,74)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Identity())

    def forward(self, x):
        b, n, _ = x.shape
        x = self.norm(x)
        q, k, v = (t.reshape(b, n, self.heads, -1).transpose(1, 2) for t in self.to_qkv(x).chunk(3, dim=-1))
        attn = self.dropout(torch.softmax((q @ k.transpose(-1, -2)) * self.scale, dim=-1))
        out = (attn @ v).transpose(1, 2).reshape(b, n, -1)
        return self.to_out(out)


class TransformerBlock(nn.Module):
    """networks/vit.py:80-96 (forward :93-96; drop_path never applied)."""

    def __init__(self, dim, heads, dim_head, mlp_dim):
        super().__init__()
        self.attn = Attention(dim, heads, dim_head)
        self.ff = FeedForward(dim, mlp_dim)

    def forward(self, x):
        x = self.attn(x) + x
        return self.ff(x) + x


class ViT(nn.Module):
    """networks/vit.py:100-139."""

    def __init__(self, image_size, image_patch_size, frames, frame_patch_size, dim, depth, heads, mlp_dim,
                 channels=1, dim_head=64):
        super().__init__()
        ih, iw = image_size if isinstance(image_size, tuple) else (image_size, image_size)
        ph = pw = image_patch_size
        assert ih % ph == 0 and iw % pw == 0, "Image dimensions must be divisible by the patch size."
        assert frames % frame_patch_size == 0, "Frames must be divisible by the frame patch size."
        self.patch = (ph, pw, frame_patch_size)
        num_patches = (ih // ph) * (iw // pw) * (frames // frame_patch_size)
        patch_dim = channels * ph * pw * frame_patch_size
        self.to_patch_embedding = nn.Sequential(_Placeholder(), nn.LayerNorm(patch_dim), nn.Linear(patch_dim, dim),
                                                nn.LayerNorm(dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches, dim))
        self.dropout = nn.Identity()  # nn.Dropout(emb_dropout = 0) (vit.py:121)
        self.transformer = nn.ModuleList([TransformerBlock(dim, heads, dim_head, mlp_dim) for _ in range(depth)])

    def patchify(self, img):
        """'b c (h p1) (w p2) (f pf) -> b (h w f) (p1 p2 pf c)'  (vit.py:115)."""
        b, c, H, W, Fr = img.shape
        p1, p2, pf = self.patch
        x = img.reshape(b, c, H // p1, p1, W // p2, p2, Fr // pf, pf).permute(0, 2, 4, 6, 3, 5, 7, 1)
        return x.reshape(b, (H // p1) * (W // p2) * (Fr // pf), p1 * p2 * pf * c)

    def forward(self, img):
        x = self.patchify(img)
        for m in list(self.to_patch_embedding)[1:]:
            x = m(x)
        x = self.dropout(x + self.pos_embedding)
        for blk in self.transformer:
            x = blk(x)
        return x


# ----------------------------------------------------------------------------------------------------------
# the three models  (networks/hybrid_CTUNet.py:694-1036)
# ----------------------------------------------------------------------------------------------------------
DIMS = [128, 256, 512, 1024]  # hybrid_CTUNet.py:727


def _check_norm(norm_name, dropout_rate=0.0):
    if norm_name != "instance":
        raise NotImplementedError("oracle restates only norm_name='instance' (the reference default)")
    if dropout_rate != 0.0:
        raise NotImplementedError("oracle restates only dropout_rate=0.0")


class _VitBranch(nn.Module):
    """Shared by CTUNet and TUNet: hybrid_CTUNet.py:732-744,786-810 / :975-1014."""

    def _build_vit_branch(self, in_channels, dim_conv_stem, out_channels, img_size, frames, patch_frame, hidden_size,
                          num_depths, mlp_dim, num_heads):
        self.patch_size = (16, 16, patch_frame)
        self.feat_size = (img_size[0] // 16, img_size[1] // 16, frames // patch_frame)
        self.hidden_size = hidden_size
        self.vit = ViT(tuple(img_size), 16, frames, patch_frame, hidden_size, num_depths, num_heads, mlp_dim)
        self.vit_encoder0 = BasicConvBlock(in_channels, dim_conv_stem, 3, 1)
        self.vit_encoder = UpAttentionBlock(hidden_size, DIMS, DS_STRIDE)
        self.vit_decoder0 = CatConvBlock(dim_conv_stem, 3)
        self.decoder_linear_96x96 = DecoderLinear(out_channels, 64)
        self.vit_out = UnetOutBlock(dim_conv_stem, out_channels)

    def proj_feat(self, x):
        b = x.shape[0]
        return x.view(b, *self.feat_size, self.hidden_size).permute(0, 4, 1, 2, 3).contiguous()  # :812-815

    def _vit_forward(self, x_in):
        d0, d1, d2 = x_in.shape[2:]
        vit_features = self.vit(x_in)
        vit_enc0 = self.vit_encoder0(x_in)
        vit_enc = self.vit_encoder(self.proj_feat(vit_features))
        vit_out = self.vit_decoder0(vit_enc[4], vit_enc0)
        vit_logits = self.vit_out(vit_out)
        b, c = vit_enc[4].shape[:2]
        tokens = vit_enc[4].reshape(b, c, -1).transpose(1, 2)
        vit_96 = self.decoder_linear_96x96(tokens, (d0, d1, d2))
        return vit_enc, vit_logits, vit_96


class CTUNet(_VitBranch):
    """networks/hybrid_CTUNet.py:694-857."""

    def __init__(self, in_channels: int, dim_conv_stem: int, out_channels: int, model_depth: int,
                 img_size: Tuple[int, int], frames: int, patch_frame: int, hidden_size: int = 768,
                 num_depths: int = 12, mlp_dim: int = 3072, num_heads: int = 12, norm_name="instance",
                 dropout_rate: float = 0.0):
        super().__init__()
        _check_norm(norm_name, dropout_rate)
        self.convnet = ResNet(model_depth, DS_STRIDE)
        self._build_vit_branch(in_channels, dim_conv_stem, out_channels, img_size, frames, patch_frame, hidden_size,
                               num_depths, mlp_dim, num_heads)
        self.res_decoder3 = Up2FusionBlock(DIMS[3], DIMS[2], 3, DS_STRIDE[3])
        self.res_decoder2 = Up2FusionBlock(DIMS[2], DIMS[1], 3, DS_STRIDE[2])
        self.res_decoder1 = Up2FusionBlock(DIMS[1], DIMS[0], 3, DS_STRIDE[1])
        self.res_decoder0 = UpConvBlock(DIMS[0], 64, 3, DS_STRIDE[0])
        self.res_out = UnetOutBlock(64, out_channels)
        self.res_out_48x48 = UnetOutBlock(DIMS[0], out_channels)
        self.res_out_24x24 = UnetOutBlock(DIMS[1], out_channels)

    def forward(self, x_in):
        vit_enc, vit_logits, vit_96 = self._vit_forward(x_in)
        e1, e2, e3, e4 = self.convnet(x_in)
        d3 = self.res_decoder3(e4, e3, vit_enc[1])
        d2 = self.res_decoder2(d3, e2, vit_enc[2])
        d1 = self.res_decoder1(d2, e1, vit_enc[3])
        out = self.res_decoder0(d1)
        return ((self.res_out(out), self.res_out_48x48(d1), self.res_out_24x24(d2)), (vit_logits, vit_96))


class CUNet(nn.Module):
    """networks/hybrid_CTUNet.py:859-937."""

    def __init__(self, out_channels: int, model_depth: int, norm_name="instance"):
        super().__init__()
        _check_norm(norm_name)
        self.convnet = ResNet(model_depth, DS_STRIDE)
        self.res_decoder3 = UpCatConvBlock(DIMS[3], DIMS[2], 3, DS_STRIDE[3])
        self.res_decoder2 = UpCatConvBlock(DIMS[2], DIMS[1], 3, DS_STRIDE[2])
        self.res_decoder1 = UpCatConvBlock(DIMS[1], DIMS[0], 3, DS_STRIDE[1])
        self.res_decoder0 = UpConvBlock(DIMS[0], 64, 3, DS_STRIDE[0])
        self.res_out = UnetOutBlock(64, out_channels)
        self.res_out_48x48 = UnetOutBlock(DIMS[0], out_channels)
        self.res_out_24x24 = UnetOutBlock(DIMS[1], out_channels)

    def forward(self, x_in):
        e1, e2, e3, e4 = self.convnet(x_in)
        d3 = self.res_decoder3(e4, e3)
        d2 = self.res_decoder2(d3, e2)
        d1 = self.res_decoder1(d2, e1)
        out = self.res_decoder0(d1)
        return (self.res_out(out), self.res_out_48x48(d1), self.res_out_24x24(d2))


class TUNet(_VitBranch):
    """networks/hybrid_CTUNet.py:939-1036."""

    def __init__(self, in_channels: int, dim_conv_stem: int, out_channels: int, img_size: Tuple[int, int],
                 frames: int, patch_frame: int, hidden_size: int = 768, num_depths: int = 12, mlp_dim: int = 3072,
                 num_heads: int = 12, norm_name="instance", dropout_rate: float = 0.0):
        super().__init__()
        _check_norm(norm_name, dropout_rate)
        self._build_vit_branch(in_channels, dim_conv_stem, out_channels, img_size, frames, patch_frame, hidden_size,
                               num_depths, mlp_dim, num_heads)

    def forward(self, x_in):
        _, vit_logits, vit_96 = self._vit_forward(x_in)
        return (vit_logits, vit_96)


# ----------------------------------------------------------------------------------------------------------
# caller contract: DiceCE loss + deep-supervision targets  (trainer_CTUNet.py:90-103, main_CTUNet.py:156-158)
# ----------------------------------------------------------------------------------------------------------
def zoom_nearest_index(n_in: int, n_out: int) -> np.ndarray:
    """scipy.ndimage.zoom(order=0, grid_mode=False) source index per output index: floor(i*(n_in-1)/(n_out-1)+0.5)
    (checked against scipy in tests)."""
    if n_out == 1:
        return np.zeros(1, dtype=np.int64)
    # scipy computes coordinate = i * zoom with zoom = (n_in-1)/(n_out-1) in double; a coordinate that rounds to
    # slightly more than n_in-1 (e.g. 48->24: 23*(47/23) = 47.00000000000001) is OUT of range and yields cval=0.
    # Such positions are returned as -1 ("constant 0").  96->48 and 96->24 (the trainer's cases) have none.
    zoom = float(n_in - 1) / float(n_out - 1)
    coord = np.arange(n_out, dtype=np.float64) * zoom
    idx = np.floor(coord + 0.5).astype(np.int64)
    idx[(coord < 0) | (coord > n_in - 1)] = -1
    return idx


def downsample_target(target: torch.Tensor, zoom: Sequence[float]) -> torch.Tensor:
    """ndimage.zoom(target, (1,1,zx,zy,zz), order=0, prefilter=False) (trainer_CTUNet.py:93-94); out = round(in*z)."""
    out = target
    for ax, z in zip((2, 3, 4), zoom):
        n_in = target.shape[ax]
        n_out = int(round(n_in * z))
        idx = torch.from_numpy(zoom_nearest_index(n_in, n_out)).to(target.device)
        sel = out.index_select(ax, idx.clamp(min=0))
        if (idx < 0).any():  # scipy's out-of-range positions read cval = 0
            shape = [1] * out.dim()
            shape[ax] = -1
            sel = sel * (idx >= 0).to(sel.dtype).view(shape)
        out = sel
    return out


def dice_ce_loss(logits: torch.Tensor, target: torch.Tensor, smooth_nr: float = 0.0, smooth_dr: float = 1e-6,
                 return_parts: bool = False):
    """MONAI 0.7.0 DiceCELoss(to_onehot_y=True, softmax=True, squared_pred=True, smooth_nr, smooth_dr)
    (main_CTUNet.py:156-158): mean over (B,C) of 1 - (2*sum(p*y)+nr)/(sum(y^2)+sum(p^2)+dr), background included,
    plus mean cross-entropy over voxels.  'parity unpinned' vs MONAI itself (not installed); restated per SURVEY 8a-H."""
    logits = logits.float()
    n_cls = logits.shape[1]
    labels = target.squeeze(1).long()
    p = torch.softmax(logits, dim=1)
    y = F.one_hot(labels, n_cls).permute(0, 4, 1, 2, 3).to(p.dtype)
    axes = (2, 3, 4)
    inter = (p * y).sum(axes)
    denom = (y * y).sum(axes) + (p * p).sum(axes)
    dice = (1.0 - (2.0 * inter + smooth_nr) / (denom + smooth_dr)).mean()
    ce = F.cross_entropy(logits, labels)
    if return_parts:
        return dice + ce, dice, ce
    return dice + ce


def ctunet_loss(outputs, target):
    """trainer_CTUNet.py:92-103."""
    (o1a, o1b, o1c), (o2a, o2b) = outputs
    t1 = downsample_target(target, (0.5, 0.5, 1.0))
    t2 = downsample_target(target, (0.25, 0.25, 0.5))
    loss1 = dice_ce_loss(o1a, target) + 0.5 * (dice_ce_loss(o1b, t1) + 0.5 * dice_ce_loss(o1c, t2))
    loss2 = dice_ce_loss(o2a, target) + dice_ce_loss(o2b, target)
    return loss1 + 0.5 * loss2


def cunet_loss(outputs, target):
    """trainer_CUNet.py:91-100."""
    o0, o1, o2 = outputs
    t1 = downsample_target(target, (0.5, 0.5, 1.0))
    t2 = downsample_target(target, (0.25, 0.25, 0.5))
    return dice_ce_loss(o0, target) + 0.5 * (dice_ce_loss(o1, t1) + 0.5 * dice_ce_loss(o2, t2))


def tunet_loss(outputs, target):
    """trainer_TUNet.py:80-82."""
    o0, o1 = outputs
    return dice_ce_loss(o0, target) + dice_ce_loss(o1, target)


# ----------------------------------------------------------------------------------------------------------
# deterministic synthetic weights / inputs (reproducible from key names alone; same on the GPU box)
# ----------------------------------------------------------------------------------------------------------
def synthetic_tensor(key: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    """uniform values keyed by crc32(key): convs/linears ~U(-1,1)/sqrt(fan_in), LN weight 1+0.1U, bias 0.1U,
    pos_embedding/rel_pos_bias 0.5U."""
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * seed) % (2 ** 31))
    u = torch.rand(tuple(shape), generator=g, dtype=torch.float32) * 2.0 - 1.0
    leaf = key.split(".")[-1]
    if leaf == "pos_embedding" or "rel_pos_bias" in key:
        return 0.5 * u
    if leaf == "bias":
        return 0.1 * u
    if len(shape) == 1:  # LayerNorm weight
        return 1.0 + 0.1 * u
    fan_in = int(np.prod(shape[1:]))
    if "transp_conv" in key:  # ConvTranspose3d weight is [Cin, Cout, k...]: each output sums over Cin only
        fan_in = int(shape[0])
    return u * (1.5 / math.sqrt(fan_in))


def synthetic_state_dict(model: nn.Module, seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: synthetic_tensor(k, v.shape, seed) for k, v in model.state_dict().items()}


def synthetic_batch(batch: int, size: Sequence[int] = (96, 96, 96), n_cls: int = 14, seed: int = 1000):
    """image U[0,1), float labels randint(0, n_cls), CPU generator seed 1000+rank."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    x = torch.rand((batch, 1, *size), generator=g, dtype=torch.float32)
    y = torch.randint(0, n_cls, (batch, 1, *size), generator=g).to(torch.float32)
    return x, y


def build(model_name: str, **kw) -> nn.Module:
    """model_name in {'ctunet','cunet','tunet'} with the BASELINE defaults (14 classes, 96^3, pf 8)."""
    model_depth = kw.pop("model_depth", 101)
    common = dict(in_channels=1, dim_conv_stem=64, out_channels=14, img_size=(96, 96), frames=96, patch_frame=8)
    common.update(kw)
    if model_name == "ctunet":
        return CTUNet(model_depth=model_depth, **common)
    if model_name == "cunet":
        return CUNet(out_channels=common["out_channels"], model_depth=model_depth)
    if model_name == "tunet":
        return TUNet(**common)
    raise ValueError(model_name)


LOSSES = {"ctunet": ctunet_loss, "cunet": cunet_loss, "tunet": tunet_loss}
