import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def pkg():
    """The product package (loads libbh.so; raises if the HIP extension is missing)."""
    import bhpkg
    return bhpkg.load()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle binding (test infrastructure)."""
    import oracle
    oracle.build()
    oracle.lib()
    return oracle


def pytest_collection_modifyitems(config, items):
    # GPU tests must never pass silently without a GPU: they are deselected by -m "not gpu";
    # if someone runs them on a box without a device they fail loudly inside bh_create.
    pass
