"""Unused names imported at networks/vit.py:7 of the reference."""


class UnetrBasicBlock:  # never instantiated by the hot path
    pass


class UnetrPrUpBlock:
    pass


class UnetrUpBlock:
    pass
