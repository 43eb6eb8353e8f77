"""Checkpoint interop with the reference.

The reference writes `{"epoch", "best_acc", "state_dict" [, "optimizer", "scheduler"]}` with `model.module.state_dict()`
under DDP (trainer_CTUNet.py:308-317) and reads it back with the "backbone." prefix stripped and strict=False
(main_CTUNet.py:166-178); `--resume_ckpt` loads a bare state dict (main_CTUNet.py:145-148).  The modules of this package
keep the reference's state_dict keys and shapes, so both directions are pure dictionary work.  Files are opened with
`torch.load(weights_only=True)`: nothing in a checkpoint is executed.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Optional, Tuple

import torch

__all__ = ["load_checkpoint", "save_checkpoint", "unwrap"]


def unwrap(model):
    """The bare network of a DistributedDataParallel / hybrid_ctunet_amd.DataParallel wrapper (both expose `.module`)."""
    return model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model


def _state_dict_of(obj) -> "OrderedDict[str, torch.Tensor]":
    sd = obj["state_dict"] if isinstance(obj, dict) and "state_dict" in obj else obj
    out = OrderedDict()
    for k, v in sd.items():
        k = k.replace("backbone.", "")          # main_CTUNet.py:171-172
        if k.startswith("module."):             # a checkpoint saved from a wrapped model
            k = k[len("module."):]
        out[k] = v
    return out


def load_checkpoint(model, checkpoint, strict: bool = False, optimizer=None) -> Tuple[int, float]:
    """Load a reference checkpoint (path or already-loaded dict; full `{"epoch","best_acc","state_dict",...}` form or a
    bare state dict) into `model` (wrapped or not).  Returns (start_epoch, best_acc) like main_CTUNet.py:173-177.
    `optimizer`, when given and present in the file, gets `load_state_dict(checkpoint["optimizer"])` - the resume the
    reference saves for but never performs."""
    if isinstance(checkpoint, (str, os.PathLike)):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=True)
    net = unwrap(model)
    missing, unexpected = net.load_state_dict(_state_dict_of(checkpoint), strict=strict)
    if strict and (missing or unexpected):
        raise RuntimeError(f"missing {missing}, unexpected {unexpected}")
    epoch = int(checkpoint.get("epoch", 0)) if isinstance(checkpoint, dict) else 0
    best = float(checkpoint.get("best_acc", 0)) if isinstance(checkpoint, dict) else 0.0
    if optimizer is not None and isinstance(checkpoint, dict) and "optimizer" in checkpoint:
        osd = checkpoint["optimizer"]
        if hasattr(optimizer, "flat") and "state" in osd:
            # torch.optim.AdamW's format (what the reference saves) into the fused optimizer: positions follow model.parameters()
            optimizer.load_state_dict(osd, params=list(net.parameters()))
        else:
            optimizer.load_state_dict(osd)
    return epoch, best


def save_checkpoint(model, epoch: int, filename: str, best_acc: float = 0, optimizer=None, scheduler=None) -> str:
    """Write the file the reference's tools read back (trainer_CTUNet.py:308-317): the unwrapped network's state_dict
    (fp32 tensors in the reference's key order), epoch, best_acc and, optionally, optimizer / scheduler state.  The fused
    optimizer's state is written in torch.optim.AdamW's format, so torch.optim.AdamW (the reference) can resume from it."""
    sd = OrderedDict((k, v.detach().cpu()) for k, v in unwrap(model).state_dict().items())
    save = {"epoch": epoch, "best_acc": best_acc, "state_dict": sd}
    if optimizer is not None:
        save["optimizer"] = optimizer.state_dict(params=list(unwrap(model).parameters())) if hasattr(optimizer, "flat") \
            else optimizer.state_dict()
    if scheduler is not None:
        save["scheduler"] = scheduler.state_dict()
    torch.save(save, filename)
    return filename
