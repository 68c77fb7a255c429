import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_abi():
    """The CPU oracle, bound through the same ctypes declarations as the product."""
    from tests import oracle_binding
    return oracle_binding.load()


@pytest.fixture(scope="session")
def hip_abi():
    from swim_amd import _lib
    return _lib.load()
