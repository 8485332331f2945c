import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "lab: opt-in kernel variants outside the product library (pytest -m lab on a GPU box)")


@pytest.fixture(scope="session")
def lib_built():
    """Builds libmvsnet_hip.so once (hipcc cross-compiles without a GPU)."""
    from mvsnet_amd.build import build_library
    return build_library(verbose=False)
