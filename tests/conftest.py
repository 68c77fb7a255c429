import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def torch_device_context_first():
    """On a GPU box, let torch create its device context BEFORE the first handle of libswimsim exists: the
    sharded tests wrap the library's buffers in torch tensors, and a lazy torch.cuda initialisation in the
    middle of a session (after the library had allocated and freed tens of GB) was measured at 9-12 minutes
    on some boxes against 1.5 s up front (profiles/r02k_pytest_gpu.log vs gpurun cycle l)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda:0")
            torch.cuda.synchronize()
    except Exception:       # noqa: BLE001 -- plumbing only; the tests that need torch will say so
        pass
    yield


@pytest.fixture(scope="session")
def oracle_abi():
    """The CPU oracle, bound through the same ctypes declarations as the product."""
    from tests import oracle_binding
    return oracle_binding.load()


@pytest.fixture(scope="session")
def hip_abi():
    from swim_amd import _lib
    return _lib.load()
