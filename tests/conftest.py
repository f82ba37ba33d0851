import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The oracle is (re)built on demand; the HIP library must already exist
    (make / __graft_entry__.build()) -- tests never build or fake it."""
    from tests import _oracle
    _oracle.build()
    yield


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
