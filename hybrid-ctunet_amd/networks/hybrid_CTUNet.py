"""CUNet / TUNet / CTUNet on MI355X kernels — host-side mirror of the reference's networks/hybrid_CTUNet.py.

Drop-in boundary: identical constructor signatures (called by keyword from main_CTUNet.py:129-143),
identical ``forward(x_in: [B,1,H,W,F]) -> tuples of [B,n_cls,...] logits`` and identical state_dict keys/shapes
(412 / 126 / 235 tensors), so reference checkpoints load with ``load_state_dict(strict=True)``.

Inside, activations are channels-last ([B, D, H, W, C]) in float32 (parity mode) or bfloat16, every op runs a
hand-written HIP kernel from libctunet_hip.so through ..ops, and the reference's einops Rearrange / torch.cat /
proj_feat layers vanish into index arithmetic (window attention addresses tokens in place; concat is a split-K
over two sources).  Returned logits are [B, n_cls, D, H, W] views of channels-last buffers padded to 16 channels.

Precision: ``model.precision`` in {"auto", "fp32", "bf16"}; "auto" follows ``torch.autocast(dtype=torch.bfloat16)``
exactly where the reference's trainer wraps ``model(data)`` in autocast (trainer_CTUNet.py:90-91), fp32 otherwise.
"""
from __future__ import annotations

from typing import Sequence, Tuple, Union

import warnings

import numpy as np
import torch
import torch.nn as nn

from .. import ops, ops_fused
from .resnet import check_norm, generate_model as resnet, get_conv_layer
from .vit import ViT, _check_dropout, feed_forward

LOGIT_PAD = ops.LOGIT_PAD  # logits are computed with N padded to 16 columns (MFMA/vector width); only the first n_cls are exposed


# ---------------------------------------------------------------------------------------------------------------
# convolutional blocks (hybrid_CTUNet.py:29-255, 593-620)
# ---------------------------------------------------------------------------------------------------------------
class ResBlock(nn.Module):
    """networks/hybrid_CTUNet.py:29-105.  conv3/norm3 are always constructed (state_dict parity) and used only when
    in_channels != out_channels or stride != 1.  forward(inp, inp2=None): inp2 is the second half of a channel concat."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, kernel_size, stride, norm_name,
                 dropout=None):
        super().__init__()
        check_norm(norm_name)
        self.conv1 = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                    dropout=dropout, conv_only=True)
        self.conv2 = get_conv_layer(spatial_dims, out_channels, out_channels, kernel_size=kernel_size, stride=1,
                                    dropout=dropout, conv_only=True)
        self.conv3 = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=1, stride=stride,
                                    dropout=dropout, conv_only=True)
        self.downsample = in_channels != out_channels
        if not np.all(np.atleast_1d(stride) == 1):
            self.downsample = True

    def forward(self, inp, inp2=None, grad_stash=None):
        if ops_fused.resblock_ok(self, inp, inp2, grad_stash):
            return ops_fused.resblock(self, inp, inp2, grad_stash)   # one autograd node replayed from launch lists
        # grad_stash: a list another consumer of `inp` parks its gradient in (ops.GradStash, created AFTER this block); it is
        # added in the data-gradient epilogue of the first convolution of this block that autograd replays.
        # identity shortcut: its gradient is folded into conv1's data-gradient epilogue (ops.GradStash) instead of an
        # autograd accumulation pass over three full-size tensors
        stash = [] if (not self.downsample and inp.requires_grad and torch.is_grad_enabled()) else None
        # conv shortcut (in != out channels, stride 1): conv3 reads the same input (pair) as conv1; its data gradients are parked
        # (GradStash on conv3's inputs) and added inside conv1's data-gradient kernel - no autograd accumulation pass over
        # three full-size tensors per input (680 MB each at vit_decoder0)
        stash1 = stash2 = None
        if (self.downsample and ops.STASH_SHORTCUT_CONV and torch.is_grad_enabled() and self.conv1.in_channels != 1
                and ops._halo_ok(self.conv1.kernel_size, self.conv1.stride, self.conv1.padding)):   # (a one-channel input is the image: no gradient flows to it)
            stash1 = [] if inp.requires_grad else None
            stash2 = [] if (inp2 is not None and inp2.requires_grad) else None
        if grad_stash is not None and stash is not None:
            raise NotImplementedError("an outside gradient stash and an identity shortcut share conv1's one residual input")
        own = stash if stash is not None else stash1
        y1 = self.conv1(inp, inp2, grad_stash=own if own is not None else grad_stash, grad_stash2=stash2)
        c2 = self.conv2
        # norm1's output feeds conv2 only: written 16-channel-blocked when conv2 runs on the halo kernel (ops.wants_b16)
        out = ops.instance_norm(y1, None, True, out_b16=ops.wants_b16(c2.conv.weight, y1, c2.stride, c2.padding))
        out = self.conv2(out)
        if self.downsample:
            # (conv3 is replayed before conv1: it takes the outside stash, its own input gradients go on to conv1)
            residual = ops.instance_norm(self.conv3(ops.GradStash.apply(inp, stash1) if stash1 is not None else inp,
                                                    ops.GradStash.apply(inp2, stash2) if stash2 is not None else inp2,
                                                    grad_stash=grad_stash if stash1 is not None else None), None, False)
        else:
            assert inp2 is None
            residual = ops.GradStash.apply(inp, stash) if stash is not None else inp
        return ops.instance_norm(out, residual, True)  # norm2 -> += residual -> LeakyReLU (:99-104)


class BasicConvBlock(nn.Module):
    """networks/hybrid_CTUNet.py:107-146."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, kernel_size, stride, norm_name):
        super().__init__()
        self.layer = ResBlock(spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name)

    def forward(self, inp):
        return self.layer(inp)


class UpCatConvBlock(nn.Module):
    """networks/hybrid_CTUNet.py:148-201: ConvT(k=s) -> cat(skip) -> ResBlock(2C -> C)."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, kernel_size, upsample_kernel_size,
                 norm_name):
        super().__init__()
        self.transp_conv = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=upsample_kernel_size,
                                          stride=upsample_kernel_size, conv_only=True, is_transposed=True)
        self.conv_block = ResBlock(spatial_dims, out_channels + out_channels, out_channels, kernel_size, 1, norm_name)

    def forward(self, inp, skip, grad_stash=None):
        return self.conv_block(self.transp_conv(inp, grad_stash=grad_stash), skip)


class UpConvBlock(nn.Module):
    """networks/hybrid_CTUNet.py:203-255."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, kernel_size, upsample_kernel_size,
                 norm_name):
        super().__init__()
        self.transp_conv = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=upsample_kernel_size,
                                          stride=upsample_kernel_size, conv_only=True, is_transposed=True)
        self.conv_block = ResBlock(spatial_dims, out_channels, out_channels, kernel_size, 1, norm_name)

    def forward(self, inp, grad_stash=None):
        return self.conv_block(self.transp_conv(inp, grad_stash=grad_stash))


class CatConvBlock(nn.Module):
    """networks/hybrid_CTUNet.py:593-620."""

    def __init__(self, spatial_dims: int, in_channels: int, kernel_size, norm_name):
        super().__init__()
        self.conv_block = ResBlock(spatial_dims, in_channels + in_channels, in_channels, kernel_size, 1, norm_name)

    def forward(self, x, skip, grad_stash=None):
        return self.conv_block(x, skip, grad_stash=grad_stash)


# ---------------------------------------------------------------------------------------------------------------
# binary cross-weight fusion (hybrid_CTUNet.py:622-669, 257-341)
# ---------------------------------------------------------------------------------------------------------------
class pixelweight_attention(nn.Module):
    """networks/hybrid_CTUNet.py:622-669.  Tokens are the voxels in natural order ('b c f h w -> b (f h w) c')."""

    def __init__(self, dim, dim_head=32, dropout=0.0):
        super().__init__()
        if _check_dropout(dropout) != 0.0:
            raise NotImplementedError("pixelweight_attention(dropout > 0): no model of the reference passes a dropout here "
                                      "(hybrid_CTUNet.py:296-297,370 build it with the default 0.0)")
        if dim_head != 32 or dim % 32 != 0:
            raise NotImplementedError("cross-weight kernel is written for dim_head = 32 (the reference's value)")
        self.dim_head = dim_head
        self.heads = dim // dim_head
        self.scale = dim_head ** -0.5
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.to_qkv1 = nn.Linear(dim, dim * 3, bias=False)
        self.to_qkv2 = nn.Linear(dim, dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(dim, dim, bias=False), nn.Dropout(dropout))

    def forward(self, x1, x2):
        if ops_fused.pwa_block_ok(self, x1, x2):
            return ops_fused.pwa_block(self, x1, x2)   # one autograd node replayed from launch lists
        qkv1 = ops.linear(ops.layer_norm(x1, self.norm1.weight, self.norm1.bias), self.to_qkv1.weight)
        qkv2 = ops.linear(ops.layer_norm(x2, self.norm2.weight, self.norm2.bias), self.to_qkv2.weight)
        out = ops.pwa(qkv1, qkv2, self.scale)
        return ops.linear(out, self.to_out[0].weight)


class Up_2Fusion_Block(nn.Module):
    """networks/hybrid_CTUNet.py:257-341; forward is the reference's 'fusion2' (:329-341)."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int, kernel_size, upsample_kernel_size,
                 norm_name):
        super().__init__()
        self.transp_conv = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=upsample_kernel_size,
                                          stride=upsample_kernel_size, conv_only=True, is_transposed=True)
        self.pixelweight_attention1 = pixelweight_attention(out_channels)
        self.pixelweight_attention2 = pixelweight_attention(out_channels)
        self.up_addconv_block1 = ResBlock(spatial_dims, out_channels, out_channels, kernel_size, 1, norm_name)
        self.up_addconv_block2 = ResBlock(spatial_dims, out_channels, out_channels, kernel_size, 1, norm_name)

    def skip_path(self, skip_conv, skip_vit):
        """First half of forward (hybrid_CTUNet.py:333-334): depends on the two encoders only, not on the decoder below."""
        return self.up_addconv_block1(self.pixelweight_attention1(skip_conv, skip_vit))

    def main_path(self, inp, skip, grad_stash=None):
        """Second half (hybrid_CTUNet.py:336-340)."""
        out = self.transp_conv(inp, grad_stash=grad_stash)
        return self.up_addconv_block2(self.pixelweight_attention2(out, skip))

    def forward(self, inp, skip_conv=None, skip_vit=None):
        if skip_vit is None:
            raise NotImplementedError("the reference's forward requires skip_vit (skip is undefined otherwise)")
        return self.main_path(inp, self.skip_path(skip_conv, skip_vit))


# ---------------------------------------------------------------------------------------------------------------
# window attention / feed-forward / pixel shuffle (hybrid_CTUNet.py:388-591)
# ---------------------------------------------------------------------------------------------------------------
class PixelShuffle(nn.Module):
    """networks/hybrid_CTUNet.py:388-432."""

    def __init__(self, spatial_dims: int, scale_factor, in_channels: int, out_channels: int):
        super().__init__()
        self.spatial_dims = spatial_dims
        self.scale_factor = tuple(scale_factor)
        self.to_out = nn.Linear(in_channels // (scale_factor[0] * scale_factor[1] * scale_factor[2]), out_channels)

    def forward(self, x):
        channels = x.shape[-1]
        f = self.scale_factor
        div = f[0] * f[1] * f[2]
        if channels % div != 0:
            raise ValueError(f"Number of input channels ({channels}) must be evenly"
                             f"divisibel by scale_factor ** dimensions ({f}**{self.spatial_dims}={div}).")
        return ops.linear(ops.pixel_shuffle(x, f), self.to_out.weight, self.to_out.bias)


class Residual(nn.Module):
    """networks/hybrid_CTUNet.py:434-440: fn(x) + x, with the add fused into fn's last GEMM epilogue."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, **kw):
        return self.fn(x, residual=x, **kw)


class MultiAxisAttention(nn.Module):
    """networks/hybrid_CTUNet.py:442-511.  forward(x, part) works on the un-partitioned channels-last volume:
    part 1 = block windows '(h h1)', part 2 = grid windows '(h1 h)' (hybrid_CTUNet.py:559,564)."""

    def __init__(self, dim, dim_head=32, dropout=0.0, window_size=7):
        super().__init__()
        _check_dropout(dropout)
        assert (dim % dim_head) == 0, 'dimension must be divisible by the head dimension'
        if dim_head not in (32, 64):
            raise NotImplementedError("attention kernels support dim_head 32 or 64")
        self.heads = dim // dim_head
        self.scale = dim_head ** -0.5
        self.window_size = window_size
        self.norm = nn.LayerNorm(dim)
        self.to_qkv = nn.Linear(dim, dim * 3, bias=False)
        self.attend = nn.Sequential(nn.Softmax(dim=-1), nn.Dropout(dropout))  # holder of p; softmax + dropout run fused
        self.to_out = nn.Sequential(nn.Linear(dim, dim, bias=False), nn.Dropout(dropout))
        self.rel_pos_bias = nn.Embedding((2 * window_size - 1) ** 3, self.heads)
        pos = torch.arange(window_size)
        grid = torch.stack(torch.meshgrid(pos, pos, pos, indexing='ij')).reshape(3, -1).t()
        rel_pos = grid[:, None, :] - grid[None, :, :] + (window_size - 1)
        m = 2 * window_size - 1
        # kept for state parity with the reference (non-persistent buffer, :479); the kernel recomputes it arithmetically
        self.register_buffer('rel_pos_indices', (rel_pos * torch.tensor([m * m, m, 1])).sum(dim=-1), persistent=False)

    def forward(self, x, residual=None, part=1):
        h, residual = ops.norm_with_residual(x, residual, self.norm.weight, self.norm.bias)
        qkv = ops.linear(h, self.to_qkv.weight)
        p = self.to_out[1].p if self.training else 0.0
        o = ops.attention(qkv, self.heads, self.scale, self.rel_pos_bias.weight, part, self.window_size, dropout_p=p)
        if p > 0.0:
            return ops.dropout(ops.linear(o, self.to_out[0].weight, None, None, 0), p, residual=residual)
        return ops.linear(o, self.to_out[0].weight, None, residual, 0)


class FeedForward(nn.Module):
    """networks/hybrid_CTUNet.py:513-526."""

    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        _check_dropout(dropout)
        inner_dim = int(dim * mult)
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(inner_dim, dim), nn.Dropout(dropout))

    def forward(self, x, residual=None):
        return feed_forward(self.net, x, residual, self.training)


class UpAttentionBlock(nn.Module):
    """networks/hybrid_CTUNet.py:528-591.  The nn.Sequential layout (Rearrange placeholders at indices 0,3,4,7) is
    kept so parameter names match: layers.<s>.0.{1,5}.fn.* (attention), {2,6}.fn.net.* (FF), 8.to_out (shuffle);
    stage 3: {1,2}.fn.net.*, 4.to_out."""

    def __init__(self, spatial_dims: int, in_channels: int, dims: tuple = (512, 256, 128, 64),
                 DS_stride: tuple = ((2, 2, 1), (2, 2, 2), (2, 2, 2), (2, 2, 2)), depth: tuple = (1, 1, 1, 1),
                 dropout: float = 0.0):
        super().__init__()
        if tuple(depth) != (1, 1, 1, 1):
            raise NotImplementedError("depth other than (1,1,1,1) is not used by the reference")
        dims = (in_channels, *dims[::-1][1:], 64)
        dim_pairs = tuple(zip(dims[:-1], dims[1:]))
        self.layers = nn.ModuleList([])
        w = 6
        self.window_size = w
        for ind, ((layer_dim_in, layer_dim), layer_depth) in enumerate(zip(dim_pairs, depth)):
            if ind <= 2:
                block = nn.Sequential(
                    nn.Identity(),
                    Residual(MultiAxisAttention(dim=layer_dim_in, dim_head=32, dropout=dropout, window_size=w)),
                    Residual(FeedForward(layer_dim_in, dropout=dropout)),
                    nn.Identity(), nn.Identity(),
                    Residual(MultiAxisAttention(dim=layer_dim_in, dim_head=32, dropout=dropout, window_size=w)),
                    Residual(FeedForward(layer_dim_in, dropout=dropout)),
                    nn.Identity(),
                    PixelShuffle(spatial_dims, DS_stride[::-1][ind], layer_dim_in, layer_dim),
                )
            else:
                block = nn.Sequential(
                    nn.Identity(),
                    Residual(FeedForward(layer_dim_in, dropout=dropout)),
                    Residual(FeedForward(layer_dim_in, dropout=dropout)),
                    nn.Identity(),
                    PixelShuffle(spatial_dims, DS_stride[::-1][ind], layer_dim_in, layer_dim),
                )
            self.layers.append(nn.Sequential(block))

    @staticmethod
    def _run_stage(blk, ind, x):
        if ops_fused.up_stage_ok(x, blk, ind, blk.training):
            return ops_fused.up_stage(blk, ind, x)   # the stage as one autograd node replayed from launch lists
        if ind <= 2:
            x = blk[1](x, part=1)   # block windows + residual
            x = blk[2](x)           # FF + residual
            x = blk[5](x, part=2)   # grid windows + residual
            x = blk[6](x)
            return blk[8](x)
        x = blk[1](x)
        x = blk[2](x)
        return blk[4](x)

    def stage_runner(self, ind):
        """Stage `ind` as a Module with forward(x) (detached from the model tree), for graphs.graph_stages."""
        return _StageRunner(self.layers[ind][0], ind)

    def forward(self, x):
        features = [x]
        for ind, stage in enumerate(self.layers):
            graphed = getattr(self, f"_graphed_stage_{ind}", None)   # graphs.graph_stages
            x = graphed(x) if graphed is not None else self._run_stage(stage[0], ind, x)
            features.append(x)
        return features


class _StageRunner(nn.Module):
    def __init__(self, blk, ind):
        super().__init__()
        self.blk, self.ind = blk, ind

    def forward(self, x):
        return UpAttentionBlock._run_stage(self.blk, self.ind, x)


class DecoderLinear(nn.Module):
    """networks/hybrid_CTUNet.py:671-691 (patch_size 1: a per-voxel Linear)."""

    def __init__(self, n_cls, patch_size, d_encoder):
        super().__init__()
        self.d_encoder = d_encoder
        self.patch_size = patch_size
        self.n_cls = n_cls
        self.head = nn.Linear(self.d_encoder, n_cls)

    def forward(self, x):
        return _head(x, self.head.weight, self.head.bias)


class UnetOutBlock(nn.Module):
    """MONAI's UnetOutBlock as the reference uses it (hybrid_CTUNet.py:781-783,810): 1x1x1 conv with bias; keys
    '<name>.conv.conv.{weight,bias}'."""

    def __init__(self, spatial_dims: int, in_channels: int, out_channels: int):
        super().__init__()
        self.conv = get_conv_layer(spatial_dims, in_channels, out_channels, kernel_size=1, stride=1, bias=True,
                                   conv_only=True)

    def forward(self, x):
        return _head(x, self.conv.conv.weight, self.conv.conv.bias)


def _head(x, weight, bias):
    """Per-voxel linear to n_cls logits, computed with N padded to LOGIT_PAD; returns a [B, n_cls, D, H, W] view."""
    return ops.head(x, weight, bias)


# ---------------------------------------------------------------------------------------------------------------
# the three models (hybrid_CTUNet.py:694-1036)
# ---------------------------------------------------------------------------------------------------------------
class _Base(nn.Module):
    precision = "auto"

    def set_precision(self, precision: str):
        if precision not in ("auto", "fp32", "bf16"):
            raise ValueError("precision must be 'auto', 'fp32' or 'bf16'")
        self.precision = precision
        return self

    _warned_fp16 = False

    def _dtype(self):
        """Compute dtype of this forward.  Under the reference's default AMP - `torch.cuda.amp.autocast()` = float16 with
        a GradScaler (trainer_CTUNet.py:21,90,106-112) - the kernels compute in bfloat16 (same operand width, fp32
        accumulation, fp32 exponent range: a scaled loss cannot overflow) and the logits are handed back as float16,
        the dtype the caller's loss expects; the GradScaler then simply never sees an inf."""
        self._fp16_out = False
        if self.precision == "bf16":
            return torch.bfloat16
        if self.precision == "fp32":
            return torch.float32
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype('cuda')
            if dt == torch.float16:
                if not _Base._warned_fp16:
                    _Base._warned_fp16 = True
                    warnings.warn("autocast(float16): the MI355X kernels compute in bfloat16 and return float16 logits; "
                                  "torch.autocast('cuda', dtype=torch.bfloat16) avoids the final cast and needs no GradScaler")
                self._fp16_out = True
            return torch.bfloat16
        return torch.float32

    def _outputs(self, outs):
        """Logits as the caller's autocast dtype expects them (float16 under the reference's default AMP)."""
        if not self._fp16_out:
            return outs
        return tuple(self._outputs(o) if isinstance(o, tuple) else o.to(torch.float16) for o in outs)

    def _input(self, x_in):
        if x_in.dim() != 5 or x_in.shape[1] != 1:
            raise ValueError(f"expected x_in of shape [B, 1, H, W, F], got {tuple(x_in.shape)}")
        if not x_in.is_cuda:
            raise RuntimeError("hybrid-ctunet_amd models run on an MI355X (HIP) device only; there is no CPU fallback")
        B, _, D, H, W = x_in.shape
        x = x_in.detach().contiguous().view(B, D, H, W, 1)
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        return ops.cast(x, self._dtype())


class _VitBranch(_Base):
    def _build_vit_branch(self, in_channels, dim_conv_stem, out_channels, img_size, frames, patch_frame, hidden_size,
                          num_depths, mlp_dim, num_heads, norm_name, dropout_rate):
        if in_channels != 1:
            raise NotImplementedError("in_channels must be 1 (the reference's ResNet stem and ViT patchify assume it)")
        if dim_conv_stem != 64:
            raise NotImplementedError("dim_conv_stem must be 64: the reference concatenates it with the 64-channel "
                                      "96^3 ViT feature (hybrid_CTUNet.py:802-810)")
        self.patch_size = (16, 16, patch_frame)
        self.feat_size = (img_size[0] // self.patch_size[0], img_size[1] // self.patch_size[1],
                          frames // self.patch_size[2])
        if any(f % 6 for f in self.feat_size):
            raise NotImplementedError(f"ViT feature grid {self.feat_size} must be divisible by the 6^3 attention window")
        self.hidden_size = hidden_size
        dims = [int(4 * item) for item in [32, 64, 128, 256]]
        DS_stride = ((2, 2, 1), (2, 2, 2), (2, 2, 2), (2, 2, 2))
        self.vit = ViT(image_size=tuple(img_size), image_patch_size=16, frames=frames, frame_patch_size=patch_frame,
                       dim=hidden_size, depth=num_depths, heads=num_heads, mlp_dim=mlp_dim, dropout=dropout_rate,
                       emb_dropout=dropout_rate, drop_path=dropout_rate)
        self.vit_encoder0 = BasicConvBlock(3, in_channels, dim_conv_stem, 3, 1, norm_name)
        self.vit_encoder = UpAttentionBlock(3, hidden_size, dims=dims, DS_stride=DS_stride, depth=(1, 1, 1, 1),
                                            dropout=dropout_rate)
        self.vit_decoder0 = CatConvBlock(3, dim_conv_stem, 3, norm_name)
        self.decoder_linear_96x96 = DecoderLinear(out_channels, 1, 64)
        self.vit_out = UnetOutBlock(3, dim_conv_stem, out_channels)
        # shape check the reference only hits at run time (torch.cat at hybrid_CTUNet.py:618): the ViT pyramid must end
        # at the input resolution
        up = (16, 16, 8)
        if tuple(f * u for f, u in zip(self.feat_size, up)) != (img_size[0], img_size[1], frames):
            raise ValueError(f"patch_frame={patch_frame} is shape-incompatible: the ViT decoder upsamples the "
                             f"{self.feat_size} grid by (16,16,8) which must equal the input size "
                             f"{(img_size[0], img_size[1], frames)} (use patch_frame=8)")

    def proj_feat(self, tokens):
        """hybrid_CTUNet.py:812-815: tokens (h w f) -> volume; channels-last makes it a pure view."""
        return tokens.view(tokens.shape[0], *self.feat_size, self.hidden_size)

    def _vit_pyramid(self, x):
        """ViT trunk + window-attention pyramid (hybrid_CTUNet.py:821,824): [768@6.6.12, 512@12.12.24, ..., 64@96^3]."""
        feats = self.vit_encoder(ops.trace_point(self.proj_feat(self.vit(x[..., 0])), "vit trunk"))
        if ops.TRACE is not None:
            feats = list(feats)
            feats[-1] = ops.trace_point(feats[-1], "window stages")
        return feats

    def _vit_heads(self, x, vit_enc, enc0=None):
        """vit_encoder0 / vit_decoder0 and the two ViT-branch heads (hybrid_CTUNet.py:822,831-835).  enc0: vit_encoder0(x) when
        the caller has already computed it (CTUNet.forward runs it on a stream of its own)."""
        # vit_enc[4] is read by vit_decoder0 and by the 96 x 96 head: the head's gradient is parked and added inside the
        # data-gradient GEMM of vit_decoder0's shortcut conv (ResBlock.forward)
        top = vit_enc[4]
        st = [] if (torch.is_grad_enabled() and top.requires_grad and ops.STASH_SHORTCUT_CONV) else None
        vit_out = self.vit_decoder0(top, enc0 if enc0 is not None else self.vit_encoder0(x), grad_stash=st)
        return self.vit_out(vit_out), self.decoder_linear_96x96(ops.GradStash.apply(top, st) if st is not None else top)

    def _vit_forward(self, x):
        vit_enc = self._vit_pyramid(x)
        vit_logits, vit_96 = self._vit_heads(x, vit_enc)
        return vit_enc, vit_logits, vit_96


class CTUNet(_VitBranch):
    """networks/hybrid_CTUNet.py:694-857."""
    overlap_branches = True   # run the ResNet and the ViT encoder on two HIP streams (set False to serialise them)
    enc0_stream = False       # vit_encoder0 on a third stream from the start of the forward pass: measured 50.1 - 53.7 ms per step
                              # against 47.1 - 47.2 without (profiles/r03_bench_ab_stream_options.log): its 96^3 kernels then
                              # share the chip with the first ResNet stages instead of filling gaps - off

    def __init__(self, in_channels: int, dim_conv_stem: int, out_channels: int, model_depth: int,
                 img_size: Tuple[int, int], frames: int, patch_frame: int, hidden_size: int = 768, num_depths: int = 12,
                 mlp_dim: int = 3072, num_heads: int = 12, norm_name: Union[Tuple, str] = "instance",
                 dropout_rate: float = 0.0) -> None:
        super().__init__()
        check_norm(norm_name)
        _check_dropout(dropout_rate)
        dims = [int(4 * item) for item in [32, 64, 128, 256]]
        DS_stride = ((2, 2, 1), (2, 2, 2), (2, 2, 2), (2, 2, 2))
        self.convnet = resnet(model_depth, DS_stride=DS_stride)
        self._build_vit_branch(in_channels, dim_conv_stem, out_channels, img_size, frames, patch_frame, hidden_size,
                               num_depths, mlp_dim, num_heads, norm_name, dropout_rate)
        self.res_decoder3 = Up_2Fusion_Block(3, dims[3], dims[2], 3, DS_stride[3], norm_name)
        self.res_decoder2 = Up_2Fusion_Block(3, dims[2], dims[1], 3, DS_stride[2], norm_name)
        self.res_decoder1 = Up_2Fusion_Block(3, dims[1], dims[0], 3, DS_stride[1], norm_name)
        self.res_decoder0 = UpConvBlock(3, dims[0], 64, 3, DS_stride[0], norm_name)
        self.res_out = UnetOutBlock(3, 64, out_channels)
        self.res_out_48x48 = UnetOutBlock(3, dims[0], out_channels)
        self.res_out_24x24 = UnetOutBlock(3, dims[1], out_channels)

    def forward(self, x_in):
        x = self._input(x_in)
        # The two encoders are independent; the reference runs the ViT branch first (hybrid_CTUNet.py:839-846).  Autograd
        # replays later-built nodes first, so with that order the ViT trunk - half of all parameters, finished within a few
        # milliseconds - would be the LAST gradients to become ready and its 350 MB all-reduce would sit exposed behind the
        # backward pass.  ResNet first: in backward the ViT branch finishes early and the long convnet backward, which
        # releases its gradients layer by layer down to the small stem, hides the communication (train.DataParallel).
        join_side = False
        # res_dec2 / res_dec1 feed the next decoder's transposed conv AND a deep-supervision head: the head's gradient is parked
        # (GradStash) and added in the epilogue of the transposed conv's data-gradient GEMM
        grad = torch.is_grad_enabled()
        stash2, stash1 = ([] if grad else None), ([] if grad else None)
        if self.overlap_branches:
            # ... and they run CONCURRENTLY, on two HIP streams: the ViT trunk (864 tokens) and the small-volume stages of both
            # branches are latency-bound launches that leave most of the 256 CUs idle; side by side they fill each other's
            # gaps and the tails of the large convolutions.  Autograd replays every node on its forward stream, so the
            # backward pass overlaps the same way.  (Workspaces are per stream, ops._wskey.)
            main = torch.cuda.current_stream()
            side = ops.side_stream(x.device)
            side.wait_stream(main)      # (recorded before the convnet is queued: the side stream starts with it, not after it)
            # vit_encoder0 (a ResBlock on the 96^3 input, hybrid_CTUNet.py:822) needs the image only: it starts at once on a third
            # stream, under the launch-latency-bound ViT trunk and the small ResNet stages, instead of behind the skip paths
            enc0 = enc0_ready = None
            if self.enc0_stream:
                third = ops.side_stream(x.device, "enc0")
                third.wait_stream(main)
                with torch.cuda.stream(third):
                    enc0 = self.vit_encoder0(x)
                    enc0.record_stream(side)
                    enc0_ready = torch.cuda.Event()
                    enc0_ready.record(third)
                x.record_stream(third)
            res_enc1, res_enc2, res_enc3, res_enc4 = self.convnet(x)
            feats_ready = torch.cuda.Event()
            feats_ready.record(main)
            with torch.cuda.stream(side):
                vit_enc = self._vit_pyramid(x)
                # The skip halves of the three fusion decoders (cross-weight fusion of the two encoders' features + a ResBlock,
                # hybrid_CTUNet.py:333-334) need the encoders only: they leave the decoder chain res_decoder3 -> 2 -> 1 and run
                # here, beside it - half of the decoders' work (res_decoder1 is the largest block of the model) off the
                # critical path, forward and backward.  The chain waits for each skip tensor where it needs it.
                side.wait_event(feats_ready)
                skips, skip_ready = [], []
                for d, e, v in ((self.res_decoder3, res_enc3, vit_enc[1]), (self.res_decoder2, res_enc2, vit_enc[2]),
                                (self.res_decoder1, res_enc1, vit_enc[3])):
                    e.record_stream(side)
                    t = ops.trace_point(d.skip_path(e, v), f"skip{3 - len(skips)}")
                    t.record_stream(main)   # allocated in the side stream's pool, consumed on the main stream
                    ev = torch.cuda.Event()
                    ev.record(side)
                    skips.append(t)
                    skip_ready.append(ev)
                if enc0_ready is not None:
                    side.wait_event(enc0_ready)
                vit_logits, vit_96x96 = self._vit_heads(x, vit_enc, enc0)   # needed by the loss only
                vit_logits = ops.trace_point(vit_logits, "vit heads")
                for t in (vit_logits, vit_96x96):
                    t.record_stream(main)
            x.record_stream(side)
            res_dec = res_enc4
            decs = []
            for d, t, ev, st in zip((self.res_decoder3, self.res_decoder2, self.res_decoder1), skips, skip_ready,
                                    (None, None, stash2)):   # res_decoder1's transposed conv reads res_dec2
                main.wait_event(ev)
                res_dec = ops.trace_point(d.main_path(res_dec, t, grad_stash=st), f"dec{3 - len(decs)}")
                decs.append(res_dec)
            res_dec3, res_dec2, res_dec1 = decs
            join_side = True
        else:
            res_enc1, res_enc2, res_enc3, res_enc4 = self.convnet(x)
            vit_enc, vit_logits, vit_96x96 = self._vit_forward(x)
            res_dec3 = self.res_decoder3(res_enc4, res_enc3, vit_enc[1])
            res_dec2 = self.res_decoder2(res_dec3, res_enc2, vit_enc[2])
            res_dec1 = self.res_decoder1.main_path(res_dec2, self.res_decoder1.skip_path(res_enc1, vit_enc[3]), grad_stash=stash2)
        park = lambda t, st: ops.GradStash.apply(t, st) if (st is not None and t.requires_grad) else t
        res_out = ops.trace_point(self.res_decoder0(res_dec1, grad_stash=stash1 if res_dec1.requires_grad else None), "dec0")
        res_logits = self.res_out(res_out)
        res_logits_48x48 = self.res_out_48x48(park(res_dec1, stash1))
        res_logits_24x24 = ops.trace_point(self.res_out_24x24(park(res_dec2, stash2)), "res heads")
        if join_side:
            torch.cuda.current_stream().wait_stream(ops.side_stream(x.device))   # the ViT-branch logits
        return self._outputs(((res_logits, res_logits_48x48, res_logits_24x24), (vit_logits, vit_96x96)))


class CUNet(_Base):
    """networks/hybrid_CTUNet.py:859-937."""

    def __init__(self, out_channels: int, model_depth: int, norm_name: Union[Tuple, str] = "instance") -> None:
        super().__init__()
        check_norm(norm_name)
        dims = [int(4 * item) for item in [32, 64, 128, 256]]
        DS_stride = ((2, 2, 1), (2, 2, 2), (2, 2, 2), (2, 2, 2))
        self.convnet = resnet(model_depth, DS_stride=DS_stride)
        self.res_decoder3 = UpCatConvBlock(3, dims[3], dims[2], 3, DS_stride[3], norm_name)
        self.res_decoder2 = UpCatConvBlock(3, dims[2], dims[1], 3, DS_stride[2], norm_name)
        self.res_decoder1 = UpCatConvBlock(3, dims[1], dims[0], 3, DS_stride[1], norm_name)
        self.res_decoder0 = UpConvBlock(3, dims[0], 64, 3, DS_stride[0], norm_name)
        self.res_out = UnetOutBlock(3, 64, out_channels)
        self.res_out_48x48 = UnetOutBlock(3, dims[0], out_channels)
        self.res_out_24x24 = UnetOutBlock(3, dims[1], out_channels)

    def forward(self, x_in):
        x = self._input(x_in)
        res_enc1, res_enc2, res_enc3, res_enc4 = self.convnet(x)
        res_dec3 = self.res_decoder3(res_enc4, res_enc3)
        grad = torch.is_grad_enabled()
        stash2, stash1 = ([] if grad else None), ([] if grad else None)   # see CTUNet.forward
        park = lambda t, st: ops.GradStash.apply(t, st) if (st is not None and t.requires_grad) else t
        res_dec2 = self.res_decoder2(res_dec3, res_enc2)
        res_dec1 = self.res_decoder1(res_dec2, res_enc1, grad_stash=stash2 if res_dec2.requires_grad else None)
        res_out = self.res_decoder0(res_dec1, grad_stash=stash1 if res_dec1.requires_grad else None)
        return self._outputs((self.res_out(res_out), self.res_out_48x48(park(res_dec1, stash1)), self.res_out_24x24(park(res_dec2, stash2))))


class TUNet(_VitBranch):
    """networks/hybrid_CTUNet.py:939-1036."""

    def __init__(self, in_channels: int, dim_conv_stem: int, out_channels: int, img_size: Tuple[int, int], frames: int,
                 patch_frame: int, hidden_size: int = 768, num_depths: int = 12, mlp_dim: int = 3072,
                 num_heads: int = 12, norm_name: Union[Tuple, str] = "instance", dropout_rate: float = 0.0) -> None:
        super().__init__()
        check_norm(norm_name)
        _check_dropout(dropout_rate)
        self._build_vit_branch(in_channels, dim_conv_stem, out_channels, img_size, frames, patch_frame, hidden_size,
                               num_depths, mlp_dim, num_heads, norm_name, dropout_rate)

    def forward(self, x_in):
        x = self._input(x_in)
        _, vit_logits, vit_96x96 = self._vit_forward(x)
        return self._outputs((vit_logits, vit_96x96))
