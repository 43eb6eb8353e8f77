"""Seeded synthetic inputs of the benchmark workload: image U[0,1) - the range
ScaleIntensityRanged(b_min=0, b_max=1) produces (utils/data_utils.py:80-82) - and float class ids randint(0, n_cls),
both from one CPU generator seeded 1000 + rank.  Same stream of numbers as the oracle's generator (a test checks it)."""
from __future__ import annotations

from typing import Sequence

import torch


def synthetic_batch(batch: int, size: Sequence[int] = (96, 96, 96), n_cls: int = 14, seed: int = 1000):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    x = torch.rand((batch, 1, *size), generator=g, dtype=torch.float32)
    y = torch.randint(0, n_cls, (batch, 1, *size), generator=g).to(torch.float32)
    return x, y
