"""Inference-side callers of `forward` on the MI355X.

`sliding_window_inference` keeps the reference's signature and semantics (trainer_CTUNet.py:417-557 for the two-output
CTUNet form, trainer_CUNet.py:268-424 for the single-output form): pad to the ROI, dense window grid with the requested
overlap, `sw_batch_size` windows per forward, predictions blended with a constant or gaussian importance map and
normalised by the accumulated weight, crop back.  The accumulation and normalisation run on the device through
`ctu_sw_accumulate` / `ctu_sw_normalize` on whatever strides the predictor returns (the models hand back channels-last
views, so nothing is permuted or copied), with one weight volume instead of a per-class count map.

`hybrid_complement` is the "Hybrid-CTUNet" output of test_CTUNet_final.py:545-551: labels of the averaged softmax of two
models' logits (and each model's own labels), one fused kernel; `dice_per_organ` the metric of trainer_CTUNet.py:49-55,293-297.

The window grid and the gaussian map follow MONAI 0.7.0's published `dense_patch_slices` / `compute_importance_map`
(the reference imports them from MONAI; see oracle/infer_oracle.py for the restatement the tests compare against).
"""
from __future__ import annotations

import math
from typing import Callable, List, Sequence, Tuple, Union

import torch
import torch.nn.functional as F

from ._lib import call, dcode, ptr, require_device, stream

__all__ = ["sliding_window_inference", "hybrid_complement", "dice_per_organ", "importance_map", "window_starts"]


def _tuple(v, n) -> Tuple:
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


def window_starts(image_size: Sequence[int], roi_size: Sequence[int], overlap: float) -> List[Tuple[int, ...]]:
    """Start corner of every window, first spatial dim outermost.  Interval = int(roi * (1 - overlap)) (roi when the image
    is exactly one window wide); the last window of a dim is shifted back inside the image."""
    per_dim = []
    for n, r in zip(image_size, roi_size):
        step = r if r == n else max(int(r * (1 - overlap)), 1)
        count = next((d for d in range(int(math.ceil(n / step))) if d * step + r >= n), None)
        count = 1 if count is None else count + 1
        per_dim.append([min(i * step, n - r) for i in range(count)])
    out = [()]
    for starts in per_dim:
        out = [o + (s,) for o in out for s in starts]
    return out


def importance_map(roi_size: Sequence[int], mode: str = "constant", sigma_scale=0.125, device=None) -> torch.Tensor:
    """Window weights.  "gaussian": a unit impulse at roi // 2 filtered by the separable erf-gaussian of sigma =
    sigma_scale * roi (truncated at 4 sigma, zero padding) and divided by its maximum - i.e. the outer product of
    g_d(x - roi_d // 2) / g_d(0) with g the unit-cell integral of the gaussian; zeros replaced by the smallest non-zero."""
    if mode == "constant":
        return torch.ones(tuple(roi_size), dtype=torch.float32, device=device)
    if mode != "gaussian":
        raise ValueError(f"blend mode must be 'constant' or 'gaussian', got {mode!r}")
    sig = _tuple(sigma_scale, len(roi_size))
    m = None
    for n, ss in zip(roi_size, sig):
        sigma = float(n * ss)
        tail = int(max(sigma * 4.0, 0.5) + 0.5)
        x = torch.arange(n, dtype=torch.float32) - (n // 2)
        t = 0.70710678 / abs(sigma)
        g = (0.5 * ((t * (x + 0.5)).erf() - (t * (x - 0.5)).erf())).clamp(min=0)
        g = torch.where(x.abs() <= tail, g, torch.zeros_like(g))  # beyond the truncated kernel's reach
        m = g if m is None else m.unsqueeze(-1) * g
    m = (m / m.max()).float()
    nz = m[m != 0].min()
    m = torch.where(m == 0, nz, m)
    return m.to(device) if device is not None else m


def sliding_window_inference(inputs: torch.Tensor, roi_size: Union[Sequence[int], int], sw_batch_size: int,
                             predictor: Callable, overlap: float = 0.25, mode: str = "constant", sigma_scale=0.125,
                             padding_mode: str = "constant", cval: float = 0.0, sw_device=None, device=None, *args,
                             outputs: str = "multi", **kwargs):
    """Drop-in for the reference's sliding_window_inference.  outputs="multi" (trainer_CTUNet.py): the predictor returns
    ((res, ...), (vit, ...)) and a tuple (blend of seg[0][0], blend of seg[1][0]) comes back; outputs="single"
    (trainer_CUNet.py / trainer_TUNet.py): the blend of predictor(...)[0].  Inputs [B, C, D, H, W] on the HIP device;
    results fp32 [B, classes, D, H, W]."""
    require_device(inputs)
    if inputs.dim() != 5:
        raise ValueError("sliding_window_inference here handles 3-D volumes: inputs [B, C, D, H, W]")
    if overlap < 0 or overlap >= 1:
        raise AssertionError("overlap must be >= 0 and < 1.")
    if outputs not in ("multi", "single"):
        raise ValueError("outputs must be 'multi' or 'single'")
    dev = inputs.device
    if (sw_device is not None and torch.device(sw_device) != dev) or (device is not None and torch.device(device) != dev):
        raise ValueError("windows and stitched outputs stay on the inputs' device (no host staging on this path)")
    image_size_ = list(inputs.shape[2:])
    B = inputs.shape[0]
    roi = _tuple(roi_size, 3)
    roi = tuple(int(r) if (r is not None and r > 0) else int(n) for r, n in zip(roi, image_size_))
    image_size = tuple(max(n, r) for n, r in zip(image_size_, roi))
    pad_size = []
    for k in range(4, 1, -1):
        diff = max(roi[k - 2] - inputs.shape[k], 0)
        pad_size.extend([diff // 2, diff - diff // 2])
    if any(pad_size):
        inputs = F.pad(inputs, pad=pad_size, mode=padding_mode, value=cval)
    starts = window_starts(image_size, roi, overlap)
    num_win = len(starts)
    total = num_win * B
    imp = importance_map(roi, mode, sigma_scale).to(dev).contiguous()
    D, H, W = image_size
    S = D * H * W
    n_out = 2 if outputs == "multi" else 1
    outs: List[torch.Tensor] = []
    count = None
    for g in range(0, total, sw_batch_size):
        idx = list(range(g, min(g + sw_batch_size, total)))
        window = torch.cat([inputs[i // num_win: i // num_win + 1, :,
                                   starts[i % num_win][0]: starts[i % num_win][0] + roi[0],
                                   starts[i % num_win][1]: starts[i % num_win][1] + roi[1],
                                   starts[i % num_win][2]: starts[i % num_win][2] + roi[2]] for i in idx])
        seg = predictor(window, *args, **kwargs)
        probs = [seg[0][0], seg[1][0]] if outputs == "multi" else [seg[0]]
        if not outs:
            C = probs[0].shape[1]
            outs = [torch.zeros((B, C, D, H, W), dtype=torch.float32, device=dev) for _ in range(n_out)]
            count = torch.zeros((B, D, H, W), dtype=torch.float32, device=dev)
        for o, p in enumerate(probs):
            require_device(p)
            if tuple(p.shape[2:]) != roi or p.shape[0] != len(idx) or p.shape[1] != outs[0].shape[1]:
                raise ValueError(f"predictor returned {tuple(p.shape)} for a window batch of {len(idx)} x {roi}")
            sn, sc, sd, sh, sw = p.stride()
            for j, i in enumerate(idx):
                d0, h0, w0 = starts[i % num_win]
                # the weight volume is shared by both outputs: accumulate it with the first one only
                call("ctu_sw_accumulate", dcode(p.dtype), p.data_ptr() + j * sn * p.element_size(), sc, sd, sh, sw, ptr(imp),
                     ptr(outs[o]), ptr(count) if o == 0 else None, outs[o].shape[1], roi[0], roi[1], roi[2], i // num_win,
                     d0, h0, w0, D, H, W, stream())
    for o in outs:
        call("ctu_sw_normalize", ptr(o), ptr(count), B, o.shape[1], S, stream())
    crop = (slice(None), slice(None),
            slice(pad_size[4], image_size_[0] + pad_size[4]), slice(pad_size[2], image_size_[1] + pad_size[2]),
            slice(pad_size[0], image_size_[2] + pad_size[0]))
    res = tuple(o[crop] for o in outs)
    return res if outputs == "multi" else res[0]


def hybrid_complement(pred1: torch.Tensor, pred2: torch.Tensor):
    """pred1, pred2: logits [C, D, H, W] of one case from two models (the CTUNet res head and the independent TUNet in
    test_CTUNet_final.py:539-551).  Returns (labels1, labels2, labels_hybrid), int64 [D, H, W]: each model's argmax and
    the argmax of the averaged softmax."""
    require_device(pred1)
    require_device(pred2)
    if pred1.shape != pred2.shape or pred1.dim() != 4:
        raise ValueError("hybrid_complement expects two [C, D, H, W] tensors of equal shape")
    p1 = pred1.float().contiguous()
    p2 = pred2.float().contiguous()
    C = p1.shape[0]
    S = p1[0].numel()
    l1, l2, lh = (torch.empty(p1.shape[1:], dtype=torch.int64, device=p1.device) for _ in range(3))
    call("ctu_hybrid_argmax", ptr(p1), ptr(p2), C, S, ptr(l1), ptr(l2), ptr(lh), stream())
    return l1, l2, lh


def dice_per_organ(pred: torch.Tensor, label: torch.Tensor, n_classes: int = 14) -> List[float]:
    """Dice of every foreground class 1 .. n_classes-1 (trainer_CTUNet.py:49-55,293-297): 2|P&L| / (|P| + |L|), 0 when the
    label has no voxel of the class."""
    out = []
    for c in range(1, n_classes):
        p, l = pred == c, label == c
        ls = int(l.sum())
        out.append(0.0 if ls == 0 else 2.0 * int((p & l).sum()) / (int(p.sum()) + ls))
    return out
