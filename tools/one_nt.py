#!/usr/bin/env python3
"""One NT GEMM shape under a profiler: python tools/one_nt.py M N K variant [nt_debug]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + sys.argv[1:]
M, N, K, variant = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dbg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
import kbench  # noqa: E402

kbench.call("ctu_set_option", b"nt_debug", dbg)
kbench.nt_model(M, N, K, variant)
