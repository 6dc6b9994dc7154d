import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc


@pytest.fixture(scope="session")
def hip():
    """The HIP library must be the thing that runs on a GPU box: fail loudly if it is missing."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from geoformer_amd import _lib

    return _lib.load()
