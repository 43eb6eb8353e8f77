"""Measurement only: `import _variant` before the package in a tool makes CTU_LIB_VARIANT=TAG load
hybrid-ctunet_amd/csrc/variants/libctunet_hip_TAG.so (tools/build_variant.sh) instead of the product library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hybrid_ctunet_amd  # noqa: E402,F401
from hybrid_ctunet_amd import _lib  # noqa: E402

tag = os.environ.get("CTU_LIB_VARIANT")
if tag:
    path = os.path.join(_lib.CSRC, "variants", f"libctunet_hip_{tag}.so")
    if not os.path.exists(path):
        raise SystemExit(f"variant library {path} is missing")
    _lib.LIB_PATH = path
    print(f"# variant library: {tag}", flush=True)
