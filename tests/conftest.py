import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a host without a GPU: gpu-marked tests are skipped, not failed."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a GPU (run with -m gpu on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure); built on demand with gcc."""
    from oracle import cpu_oracle
    cpu_oracle.build()
    return cpu_oracle


@pytest.fixture(scope="session")
def mc():
    """The product package; importing it loads libmcadcensus.so or raises."""
    import mc_cnn_amd
    return mc_cnn_amd
