"""3-D ViT encoder on MI355X kernels — host-side mirror of the reference's networks/vit.py.

Same constructors, module tree and state_dict keys (``to_patch_embedding.{1,2,3}``, ``pos_embedding``,
``transformer.<i>.attn.{norm,to_qkv,to_out.0}``, ``transformer.<i>.ff.net.{0,1,4}``).  nn.Linear / nn.LayerNorm are
parameter holders only; arithmetic runs through ..ops (LayerNorm, MFMA GEMMs with fused bias/GELU/residual
epilogues, fused softmax attention).
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops, ops_fused


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


def _check_dropout(p):
    """nn.Dropout's own argument check (0 <= p <= 1) narrowed to p < 1 (p = 1 zeroes everything: not a training recipe)."""
    if not 0.0 <= p < 1.0:
        raise ValueError(f"dropout probability has to be in [0, 1), got {p}")
    return float(p)


class FeedForward(nn.Module):
    """networks/vit.py:31-44: LayerNorm -> Linear -> GELU(erf) -> Linear.  forward(x, residual) fuses `+ x`."""

    def __init__(self, dim, hidden_dim, dropout=0.0):
        super().__init__()
        _check_dropout(dropout)
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))

    def forward(self, x, residual=None):
        return feed_forward(self.net, x, residual, self.training)


def feed_forward(n, x, residual, training):
    """LayerNorm -> Linear -> GELU -> Dropout -> Linear -> Dropout (+ residual) on the parameter holders of `n`.  The
    dropout-free case keeps the residual in the second GEMM's epilogue."""
    p1, p2 = (n[3].p, n[5].p) if training else (0.0, 0.0)
    h, residual = ops.norm_with_residual(x, residual, n[0].weight, n[0].bias)
    h = ops.linear(h, n[1].weight, n[1].bias, None, 1)
    if p1 > 0.0:
        h = ops.dropout(h, p1)
    if p2 > 0.0:
        return ops.dropout(ops.linear(h, n[4].weight, n[4].bias, None, 0), p2, residual=residual)
    return ops.linear(h, n[4].weight, n[4].bias, residual, 0)


class Attention(nn.Module):
    """networks/vit.py:46-78: pre-LN inside, qkv without bias, softmax(q k^T * scale) v, out-projection with bias."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        _check_dropout(dropout)
        inner_dim = dim_head * heads
        project_out = not (heads == 1 and dim_head == dim)
        if not project_out:
            raise NotImplementedError("heads == 1 and dim_head == dim (no out-projection) is not used by the reference")
        if dim_head not in (32, 64):
            raise NotImplementedError("attention kernels support dim_head 32 or 64")
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.dropout = nn.Dropout(dropout)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))

    def forward(self, x, residual=None):
        p = self.dropout.p if self.training else 0.0
        h, residual = ops.norm_with_residual(x, residual, self.norm.weight, self.norm.bias)
        qkv = ops.linear(h, self.to_qkv.weight)
        o = ops.attention(qkv, self.heads, self.scale, dropout_p=p)
        if p > 0.0:
            return ops.dropout(ops.linear(o, self.to_out[0].weight, self.to_out[0].bias, None, 0), p, residual=residual)
        return ops.linear(o, self.to_out[0].weight, self.to_out[0].bias, residual, 0)


class TransformerBlock(nn.Module):
    """networks/vit.py:80-96 (forward :93-96: x = attn(x) + x; x = ff(x) + x; drop_path is never applied)."""

    def __init__(self, dim, heads, dim_head, mlp_dim, dropout=0.0, drop_path=0.0):
        super().__init__()
        self.attn = Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout)
        self.ff = FeedForward(dim, mlp_dim, dropout=dropout)
        self.drop_path = nn.Identity()

    def forward(self, x):
        x = self.attn(x, residual=x)
        return self.ff(x, residual=x)


class ViT(nn.Module):
    """networks/vit.py:100-139.  forward(img) takes the single-channel volume as [B, H, W, F] (== [B,1,H,W,F]) and
    returns tokens [B, (h w f), dim]."""

    def __init__(self, image_size, image_patch_size, frames, frame_patch_size, dim, depth, heads, mlp_dim, channels=1,
                 dim_head=64, dropout=0.0, emb_dropout=0.0, drop_path=0.0):
        super().__init__()
        _check_dropout(dropout)
        _check_dropout(emb_dropout)
        image_height, image_width = pair(image_size)
        patch_height, patch_width = pair(image_patch_size)
        assert image_height % patch_height == 0 and image_width % patch_width == 0, \
            'Image dimensions must be divisible by the patch size.'
        assert frames % frame_patch_size == 0, 'Frames must be divisible by the frame patch size.'
        if channels != 1:
            raise NotImplementedError("patchify kernel handles the single-channel volumes the reference feeds")
        self.patch = (patch_height, patch_width, frame_patch_size)
        num_patches = (image_height // patch_height) * (image_width // patch_width) * (frames // frame_patch_size)
        patch_dim = channels * patch_height * patch_width * frame_patch_size
        self.to_patch_embedding = nn.Sequential(nn.Identity(), nn.LayerNorm(patch_dim), nn.Linear(patch_dim, dim),
                                                nn.LayerNorm(dim))
        self.pos_embedding = nn.Parameter(torch.randn(1, num_patches, dim))
        self.dropout = nn.Dropout(emb_dropout)
        self.transformer = nn.ModuleList(
            [TransformerBlock(dim, heads, dim_head, mlp_dim, dropout, drop_path) for _ in range(depth)])

    def forward(self, img):
        e = self.to_patch_embedding
        x = ops.patchify(img, *self.patch)  # 'b c (h p1)(w p2)(f pf) -> b (h w f)(p1 p2 pf c)', vit.py:115
        x = ops.layer_norm(x, e[1].weight, e[1].bias)
        x = ops.linear(x, e[2].weight, e[2].bias)
        x = ops.layer_norm(x, e[3].weight, e[3].bias)
        x = ops.add_bcast(x, self.pos_embedding)
        x = ops.dropout(x, self.dropout.p, self.training)
        graphed = getattr(self, "_graphed_transformer", None)   # graphs.graph_stages: the 12 blocks as two graph launches
        if graphed is not None:
            return graphed(x)
        if ops_fused.vit_trunk_ok(self, x):
            return ops_fused.vit_trunk(self.transformer, x)   # the 12 blocks as one autograd node replayed from launch lists
        for blk in self.transformer:
            x = blk(x)
        return x
