import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_gpu_cache(request):
    """After every GPU test hand torch's cached device blocks back, so one test's high-water mark (the fp32 whole-model
    runs hold tens of GB) does not shape the next test's allocations."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc

        import torch
        if torch.cuda.is_available():
            gc.collect()
            torch.cuda.empty_cache()
