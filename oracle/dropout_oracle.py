"""TEST INFRASTRUCTURE ONLY - CPU restatement of the dropout path.

The reference applies torch's nn.Dropout (networks/vit.py:38,40,57,63,74; networks/hybrid_CTUNet.py:459-467,521-523):
which elements fall is torch's CUDA Philox stream, an implementation detail no port can (or should) reproduce.  What IS
checkable, and what this module pins:
  * the generator: Philox4x32-10 restated in numpy and checked against the Random123 known-answer vectors
    (tests/test_dropout_cpu.py) - the device code in csrc/philox.h must produce the same bits;
  * the mask layout (which counter / draw belongs to which element) - restated here independently of the kernels;
  * the arithmetic GIVEN a mask: `install()` puts mask-driven stand-ins at exactly the places where the reference has
    nn.Dropout modules inside oracle/ctunet_oracle.py's restated blocks, so outputs and gradients of the HIP path can be
    compared with the reference's own formulae under the very same masks.
PARITY NOTE: the identity of the dropped elements is "parity unpinned" by construction (stochastic op); distributional
properties (keep rate, 1/(1-p) scaling, eval = identity) are tested instead.

Only tests/ may import this module.
"""
import numpy as np
import torch
from torch import nn

from . import ctunet_oracle as O

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
U32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11) on uint32 arrays (broadcast); returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & U32 for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & U32, p1 >> np.uint64(32), p1 & U32
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def thr16(p: float) -> int:
    """Keep iff the 16-bit draw >= thr16; float32 arithmetic like the device (csrc/dropout.hip: ctu_make_drop_ctx)."""
    return min(int(np.float32(p) * np.float32(65536.0) + np.float32(0.5)), 65535)


def scale(p: float) -> float:
    return float(np.float32(65536.0) / np.float32(65536 - thr16(p)))


def _draw(words, j):
    """draw j (0..7) of a call: bits [16 (j & 1), +16) of output word j >> 1."""
    w = np.choose(j >> 1, words)
    return (w >> (16 * (j & 1)).astype(np.uint32)) & np.uint32(0xFFFF)


def flat_keep(n: int, p: float, seed: int, offset: int) -> np.ndarray:
    """keep flags of a flat tensor of n elements: element i -> counter (i >> 3, i >> 35, 0xD0D0D0D0, offset), draw i & 7."""
    i = np.arange(n, dtype=np.uint64)
    words = philox4x32_10(i >> np.uint64(3), i >> np.uint64(35), 0xD0D0D0D0, offset & 0xFFFFFFFF, seed & 0xFFFFFFFF, seed >> 32)
    return _draw(words, (i & np.uint64(7)).astype(np.int64)) >= thr16(p)


def attn_keep(pairs: int, ntok: int, p: float, seed: int, offset: int) -> np.ndarray:
    """keep[pair][query][key]: a call covers 2 queries x 4 keys - counter (key >> 2, query >> 1, pair, offset),
    draw 4 (query & 1) + (key & 3)."""
    pr, q, k = np.meshgrid(np.arange(pairs), np.arange(ntok), np.arange(ntok), indexing="ij")
    words = philox4x32_10(k >> 2, q >> 1, pr, offset & 0xFFFFFFFF, seed & 0xFFFFFFFF, seed >> 32)
    return _draw(words, 4 * (q & 1) + (k & 3)) >= thr16(p)


class Provider:
    """Hands out masks in dropout-call order, like hybrid_ctunet_amd.ops does: call i of a step uses offset base + i."""

    def __init__(self, p: float, seed: int, offset: int = 0):
        self.p, self.seed, self.offset = p, seed, offset

    def _take(self):
        o = self.offset
        self.offset += 1
        return o

    def flat(self, shape):
        n = int(np.prod(shape))
        return torch.from_numpy(flat_keep(n, self.p, self.seed, self._take()).reshape(tuple(shape)))

    def attn(self, pairs, ntok):
        return torch.from_numpy(attn_keep(pairs, ntok, self.p, self.seed, self._take()))


class MaskedDropout(nn.Module):
    """Stands where the reference has nn.Dropout(p) on activations.  `mode`: None - the tensor is in the product's layout
    already; "block"/"grid" - the tensor is the window-partitioned view (b, X, Y, Z, w1, w2, w3, c) of a channels-last
    volume, whose flat order defines the mask on the device."""

    def __init__(self, provider: Provider, mode=None):
        super().__init__()
        self.provider, self.mode = provider, mode

    def forward(self, x):
        if self.mode is None:
            keep = self.provider.flat(x.shape)
        else:
            # (MultiAxisAttention flattens the windows to (b X Y Z, w1, w2, w3, c) before to_out: `vol` remembers b, X, Y, Z)
            b, X, Y, Z, w, _, _, c = x.shape if x.dim() == 8 else self.vol
            vol = self.provider.flat((b, X * w, Y * w, Z * w, c))                       # [B, D, H, W, C]
            keep = O._partition(vol.permute(0, 4, 1, 2, 3), w, self.mode).reshape(x.shape)
        return x * keep.to(x.dtype) * scale(self.provider.p)


class MaskedAttnDropout(nn.Module):
    """Stands where the reference drops attention probabilities: attn (G, heads, n, n)."""

    def __init__(self, provider: Provider):
        super().__init__()
        self.provider = provider

    def forward(self, attn):
        G, h, n, _ = attn.shape
        keep = self.provider.attn(G * h, n).reshape(G, h, n, n)
        return attn * keep.to(attn.dtype) * scale(self.provider.p)


def install(module: nn.Module, provider: Provider, mode=None) -> nn.Module:
    """Replace the Identity stand-ins of nn.Dropout inside an oracle block by mask-driven ones (in place)."""
    if isinstance(module, O.FeedForward):
        module.net[3] = MaskedDropout(provider, mode)
        module.net[5] = MaskedDropout(provider, mode)
    elif isinstance(module, O.Attention):
        module.dropout = MaskedAttnDropout(provider)
        module.to_out[1] = MaskedDropout(provider)
    elif isinstance(module, O.MultiAxisAttention):
        module.attend[1] = MaskedAttnDropout(provider)
        module.to_out[1] = drop = MaskedDropout(provider, mode)
        module.register_forward_pre_hook(lambda m, inp: setattr(drop, "vol", tuple(inp[0].shape)))
    elif isinstance(module, O.TransformerBlock):
        install(module.attn, provider)
        install(module.ff, provider)
    elif isinstance(module, O.ViT):
        module.dropout = MaskedDropout(provider)
        for blk in module.transformer:
            install(blk, provider)
    elif isinstance(module, O.Residual):
        install(module.fn, provider, mode)
    elif isinstance(module, O.UpAttentionBlock):
        for ind, layer in enumerate(module.layers):
            blk = layer[0]
            if ind <= 2:
                for i, m in ((1, "block"), (2, "block"), (5, "grid"), (6, "grid")):
                    install(blk[i], provider, m)
            else:
                install(blk[1], provider)
                install(blk[2], provider)
    else:
        raise TypeError(f"no dropout sites known for {type(module).__name__}")
    return module
