"""TEST-ONLY stand-in for the six MONAI 0.7.0 wrapper symbols the reference's hot path imports.

MONAI (requirements.txt:1 of the reference, pinned 0.7.0) is not installed in this image and cannot be
fetched (no network). None of the six symbols contains arithmetic: they only construct torch.nn leaf modules.
This package restates their construction semantics so that the reference's own networks/*.py can be imported
UNCHANGED by tests/golden/make_golden.py to generate golden vectors.  It is never imported by the product
package, by bench.py or by anything that runs on the GPU box.

PARITY CAVEAT ("parity unpinned" at the MONAI boundary): the semantics below are restated
from knowledge of the MONAI 0.7.0 release, not from its source.  Load-bearing assumptions:
  (i)  Norm.INSTANCE -> torch.nn.InstanceNorm3d(num_features) with torch defaults (eps 1e-5, affine=False,
       no running stats);
  (ii) Convolution(conv_only=True) registers exactly one child module named "conv".
"""
