#!/usr/bin/env python3
"""bench.py on a measurement variant of the library: CTU_LIB_VARIANT=TAG python tools/bench_variant.py <bench.py arguments>."""
import os
import sys

import _variant  # noqa: F401  (sets _lib.LIB_PATH before the first call loads the library)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

bench.main()
