import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip():
    """GPU tests call the product through the C ABI; a missing library or GPU is a FAILURE, not a skip."""
    import torch
    from motionrag_amd import _lib
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    _lib.build()
    return _lib.lib()
