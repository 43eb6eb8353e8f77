"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's inference-side callers of `forward` (SURVEY.md 8f rows
1-3): gaussian-blended sliding-window inference with two accumulated outputs, the single-output variant, the hybrid
softmax-average/argmax complementation with per-organ Dice, and the checkpoint dictionary format.

Only tests/ may import this module; the product (hybrid-ctunet_amd/inference.py, checkpoint.py) never does.

Reference (read as text, /root/reference):
  * trainer_CTUNet.py:417-581  sliding_window_inference (two outputs: seg_prob[0][0] and seg_prob[1][0]) and _get_scan_interval
  * trainer_CUNet.py:268-424   the single-output variant (predictor(...)[0])
  * test_CTUNet_final.py:539-551, trainer_CTUNet.py:283-299  softmax -> average -> argmax, dice per organ 1..13
  * trainer_CTUNet.py:308-317 (save_checkpoint), main_CTUNet.py:166-178 (load: "backbone." stripped, strict=False)

PARITY UNPINNED AT THE MONAI BOUNDARY: the reference calls four MONAI 0.7.0 helpers that are not installed here and not
vendored in /root/reference - `dense_patch_slices`, `get_valid_patch_size`, `fall_back_tuple` (monai/data/utils.py,
monai/utils/misc.py) and `compute_importance_map` (monai/data/utils.py, built on `GaussianFilter` /
`gaussian_1d(approx="erf", truncated=4.0)` of monai/networks/layers).  They are restated below from MONAI 0.7.0's
published algorithm; everything around them follows the reference's own lines.  No golden vector exists for this row
(the reference has no test for it and MONAI cannot be imported), so the tests pin it through closed forms and
size-independent properties instead (partition of unity, window coverage, translation of a known predictor).
"""
import math
from typing import Callable, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ---- MONAI 0.7.0 helpers (restated) -------------------------------------------------------------------------------
def fall_back_tuple(user, default) -> Tuple[int, ...]:
    """monai.utils.misc.fall_back_tuple: non-positive / None components fall back to `default`."""
    if isinstance(user, int):
        user = (user,) * len(default)
    return tuple(int(u) if (u is not None and u > 0) else int(d) for u, d in zip(user, default))


def get_valid_patch_size(image_size: Sequence[int], patch_size: Sequence[int]) -> Tuple[int, ...]:
    """monai.data.utils.get_valid_patch_size: min(patch, image) per dim after fall-back."""
    ps = fall_back_tuple(patch_size, image_size)
    return tuple(min(p, i) for p, i in zip(ps, image_size))


def dense_patch_slices(image_size: Sequence[int], patch_size: Sequence[int], scan_interval: Sequence[int]):
    """monai.data.utils.dense_patch_slices (0.7.0): per dim the window starts i*interval, the last one shifted back inside
    the image; first spatial dim outermost ("ij" meshgrid order)."""
    nd = len(image_size)
    patch_size = get_valid_patch_size(image_size, patch_size)
    scan_num = []
    for i in range(nd):
        if scan_interval[i] == 0:
            scan_num.append(1)
        else:
            num = int(math.ceil(float(image_size[i]) / scan_interval[i]))
            scan_dim = next((d for d in range(num) if d * scan_interval[i] + patch_size[i] >= image_size[i]), None)
            scan_num.append(scan_dim + 1 if scan_dim is not None else 1)
    starts = []
    for dim in range(nd):
        ds = []
        for idx in range(scan_num[dim]):
            s = idx * scan_interval[dim]
            s -= max(s + patch_size[dim] - image_size[dim], 0)
            ds.append(s)
        starts.append(ds)
    out = np.asarray([x.flatten() for x in np.meshgrid(*starts, indexing="ij")]).T
    return [tuple(slice(int(s), int(s) + patch_size[d]) for d, s in enumerate(x)) for x in out]


def gaussian_1d(sigma: float, truncated: float = 4.0) -> torch.Tensor:
    """monai.networks.layers.convutils.gaussian_1d, approx="erf": integral of the gaussian over each unit cell."""
    tail = int(max(float(sigma) * truncated, 0.5) + 0.5)
    x = torch.arange(-tail, tail + 1, dtype=torch.float)
    t = 0.70710678 / abs(float(sigma))
    out = 0.5 * ((t * (x + 0.5)).erf() - (t * (x - 0.5)).erf())
    return out.clamp(min=0)


def compute_importance_map(patch_size: Sequence[int], mode: str = "constant", sigma_scale=0.125) -> torch.Tensor:
    """monai.data.utils.compute_importance_map (0.7.0).  gaussian: a unit impulse at the patch centre (i // 2) filtered
    by the separable erf-gaussian (zero padding), divided by its maximum, zeros replaced by the smallest non-zero."""
    if mode == "constant":
        return torch.ones(tuple(patch_size), dtype=torch.float)
    if mode != "gaussian":
        raise ValueError(f"unsupported blend mode {mode}")
    if not isinstance(sigma_scale, (tuple, list)):
        sigma_scale = (sigma_scale,) * len(patch_size)
    m = torch.zeros(tuple(patch_size), dtype=torch.float)
    m[tuple(i // 2 for i in patch_size)] = 1
    x = m[None, None]
    nd = len(patch_size)
    for d, (n, ss) in enumerate(zip(patch_size, sigma_scale)):
        k = gaussian_1d(n * ss)
        shape = [1, 1] + [1] * nd
        shape[2 + d] = k.numel()
        pad = [0, 0] * nd
        pad[2 * (nd - 1 - d)] = pad[2 * (nd - 1 - d) + 1] = k.numel() // 2
        conv = (F.conv1d, F.conv2d, F.conv3d)[nd - 1]
        x = conv(F.pad(x, pad), k.reshape(shape))
    m = x[0, 0]
    m = (m / m.max()).float()
    nz = m[m != 0].min().item()
    m[m == 0] = nz
    return m


# ---- the reference's own code ------------------------------------------------------------------------------------
def get_scan_interval(image_size, roi_size, overlap: float) -> Tuple[int, ...]:
    """trainer_CTUNet.py:560-581."""
    out = []
    for i in range(len(roi_size)):
        if roi_size[i] == image_size[i]:
            out.append(int(roi_size[i]))
        else:
            iv = int(roi_size[i] * (1 - overlap))
            out.append(iv if iv > 0 else 1)
    return tuple(out)


def sliding_window_inference(inputs: torch.Tensor, roi_size, sw_batch_size: int, predictor: Callable, overlap: float = 0.25,
                             mode: str = "constant", sigma_scale=0.125, padding_mode: str = "constant", cval: float = 0.0,
                             outputs: str = "multi"):
    """trainer_CTUNet.py:417-557 (outputs="multi": seg_prob[0][0] and seg_prob[1][0] accumulated separately, a tuple is
    returned) and trainer_CUNet.py:268-424 (outputs="single": predictor(...)[0], one tensor)."""
    nd = inputs.dim() - 2
    if overlap < 0 or overlap >= 1:
        raise AssertionError("overlap must be >= 0 and < 1.")
    image_size_ = list(inputs.shape[2:])
    batch_size = inputs.shape[0]
    roi_size = fall_back_tuple(roi_size, image_size_)
    image_size = tuple(max(image_size_[i], roi_size[i]) for i in range(nd))
    pad_size = []
    for k in range(inputs.dim() - 1, 1, -1):
        diff = max(roi_size[k - 2] - inputs.shape[k], 0)
        half = diff // 2
        pad_size.extend([half, diff - half])
    inputs = F.pad(inputs, pad=pad_size, mode=padding_mode, value=cval)
    scan_interval = get_scan_interval(image_size, roi_size, overlap)
    slices = dense_patch_slices(image_size, roi_size, scan_interval)
    num_win = len(slices)
    total = num_win * batch_size
    imp = compute_importance_map(get_valid_patch_size(image_size, roi_size), mode=mode, sigma_scale=sigma_scale)
    n_out = 2 if outputs == "multi" else 1
    out_img: List[torch.Tensor] = []
    cnt: List[torch.Tensor] = []
    for g in range(0, total, sw_batch_size):
        rng = range(g, min(g + sw_batch_size, total))
        unravel = [[slice(int(i / num_win), int(i / num_win) + 1), slice(None)] + list(slices[i % num_win]) for i in rng]
        window = torch.cat([inputs[tuple(w)] for w in unravel])
        seg = predictor(window)
        probs = [seg[0][0], seg[1][0]] if outputs == "multi" else [seg[0]]
        if not out_img:
            shape = [batch_size, probs[0].shape[1]] + list(image_size)
            out_img = [torch.zeros(shape, dtype=torch.float32) for _ in range(n_out)]
            cnt = [torch.zeros(shape, dtype=torch.float32) for _ in range(n_out)]
        for i, w in zip(rng, unravel):
            for o in range(n_out):
                out_img[o][tuple(w)] += imp * probs[o][i - g].float()
                cnt[o][tuple(w)] += imp
    out_img = [o / c for o, c in zip(out_img, cnt)]
    final = []
    for sp in range(nd):
        final.insert(0, slice(pad_size[sp * 2], image_size_[nd - sp - 1] + pad_size[sp * 2]))
    while len(final) < out_img[0].dim():
        final.insert(0, slice(None))
    res = tuple(o[tuple(final)] for o in out_img)
    return res if outputs == "multi" else res[0]


def hybrid_complement(pred1: torch.Tensor, pred2: torch.Tensor):
    """test_CTUNet_final.py:545-551 on [C, D, H, W] logits: per-model argmax and the argmax of the averaged softmax."""
    s1, s2 = torch.softmax(pred1.float(), 0), torch.softmax(pred2.float(), 0)
    return torch.argmax(s1, 0), torch.argmax(s2, 0), torch.argmax((s1 + s2) / 2.0, 0)


def dice(x: np.ndarray, y: np.ndarray) -> float:
    """trainer_CTUNet.py:49-55 (`dice`): 2|x&y| / (|x| + |y|), 0 when y is empty."""
    inter = np.sum(np.sum(np.sum(x * y)))
    y_sum = np.sum(np.sum(np.sum(y)))
    if y_sum == 0:
        return 0.0
    x_sum = np.sum(np.sum(np.sum(x)))
    return 2 * inter / (x_sum + y_sum)
