#!/usr/bin/env python3
"""One halo conv shape under a profiler: python tools/one_halo.py B D H W C N [nt_debug]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B, D, H, W, C, N = (int(v) for v in sys.argv[1:7])
dbg = int(sys.argv[7]) if len(sys.argv) > 7 else 0
import kbench  # noqa: E402

kbench.call("ctu_set_option", b"nt_debug", dbg)
kbench.halo(B, D, H, W, C, N, "fwd")
