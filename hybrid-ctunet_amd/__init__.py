"""hybrid-ctunet_amd — MI355X-native (gfx950) implementation of Hybrid-CTUNet's volumetric forward/backward hot path.

The directory name carries a hyphen (it mirrors the reference repository's name); import it as ``hybrid_ctunet_amd``
through the shim module at the repository root, which registers this directory under that name.

    from hybrid_ctunet_amd.networks.hybrid_CTUNet import CTUNet, CUNet, TUNet   # drop-in for the reference's import

Nothing here falls back to eager PyTorch or to the CPU: the kernels live in csrc/libctunet_hip.so (built by
``__graft_entry__.build()``), and every op raises if that library or a HIP device is missing.
"""
import os as _os

# The training step runs on five to six HIP streams (two encoder branches, weight-gradient companions, the gradient-exchange
# stream) and RCCL adds its own.  ROCclr maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and two of the step's
# streams sharing one queue cost 8 ms of a 47 ms step (bench.py, profiles/r03_bench_rccl_group_vs_hw_queues.log).  Read once when
# HIP initialises, so this only helps when the package is imported before the first GPU call; a value the caller set stays.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .networks.hybrid_CTUNet import CTUNet, CUNet, TUNet  # noqa: F401,E402
from .train import (DataParallel, FlatParams, FusedAdamW, GraphedStep, LOSSES, ctunet_loss, cunet_loss, dice_ce_loss,  # noqa: F401
                    gradient_ready_order, tunet_loss)
from .inference import dice_per_organ, hybrid_complement, sliding_window_inference  # noqa: F401
from .checkpoint import load_checkpoint, save_checkpoint  # noqa: F401
from .synthetic import synthetic_batch  # noqa: F401
from .graphs import graph_stages  # noqa: F401

__all__ = ["CTUNet", "CUNet", "TUNet", "DataParallel", "FlatParams", "FusedAdamW", "GraphedStep", "LOSSES", "ctunet_loss",
           "cunet_loss", "tunet_loss", "dice_ce_loss", "gradient_ready_order", "sliding_window_inference",
           "hybrid_complement", "dice_per_organ", "load_checkpoint", "save_checkpoint", "synthetic_batch"]


def build_model(name: str, model_depth: int = 101, **kw):
    """'ctunet' | 'cunet' | 'tunet' with the BASELINE.json configuration (14 classes, 96^3 ROI, patch_frame 8)."""
    common = dict(in_channels=1, dim_conv_stem=64, out_channels=14, img_size=(96, 96), frames=96, patch_frame=8)
    common.update(kw)
    if name == "ctunet":
        return CTUNet(model_depth=model_depth, **common)
    if name == "cunet":
        return CUNet(out_channels=common["out_channels"], model_depth=model_depth)
    if name == "tunet":
        return TUNet(**common)
    raise ValueError(name)
