"""Stand-in for monai.networks.layers.utils.{get_act_layer,get_norm_layer}
(used at networks/resnet.py:12,97-102,156-157,198 and networks/hybrid_CTUNet.py:20,84-87)."""
import torch.nn as nn


def get_norm_layer(name, spatial_dims=1, channels=1):
    if isinstance(name, (tuple, list)):
        name, kwargs = name[0], dict(name[1])
    else:
        kwargs = {}
    name = str(name).lower()
    if name == "instance" and spatial_dims == 3:
        # MONAI passes only num_features -> torch defaults: eps=1e-5, affine=False, track_running_stats=False
        return nn.InstanceNorm3d(channels, **kwargs)
    raise NotImplementedError(f"stand-in supports only instance norm in 3-D, got {name!r}/{spatial_dims}")


def get_act_layer(name):
    if isinstance(name, (tuple, list)):
        name, kwargs = name[0], dict(name[1])
    else:
        kwargs = {}
    name = str(name).lower()
    if name == "leakyrelu":
        return nn.LeakyReLU(**kwargs)
    raise NotImplementedError(f"stand-in supports only leakyrelu, got {name!r}")
