import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name has a hyphen, so it is imported by string)."""
    return importlib.import_module("ascii-renderer_amd")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle: the checker, never the thing under test in the parity tests."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
