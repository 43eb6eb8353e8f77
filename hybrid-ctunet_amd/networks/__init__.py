"""Host-side mirror of the reference's ``networks`` package (hybrid_CTUNet.py, resnet.py, vit.py)."""
