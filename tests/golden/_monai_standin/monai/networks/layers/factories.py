"""Stand-in for monai.networks.layers.factories.{Act,Norm}: plain string constants (used by
networks/resnet.py:11,23-24,90 of the reference only as default-argument values)."""


class Act:
    PRELU = "prelu"
    RELU = "relu"
    LEAKYRELU = "leakyrelu"
    GELU = "gelu"


class Norm:
    INSTANCE = "instance"
    BATCH = "batch"
    GROUP = "group"
    LAYER = "layer"
