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


@pytest.fixture(scope="session")
def ref():
    """The reference's own kernels (oracle/_ref: /root/reference/adcensus.cu compiled for gfx950; test infrastructure).
    Built only where /root/reference exists and shipped to the GPU box as a prebuilt .so.  Without it the parity story
    would silently fall back to oracle-only, so: MC_REQUIRE_REF=1 (scripts/gpu_tests.sh, scripts/gpu_evidence.sh) turns
    the skip into a failure."""
    from oracle.ref_lib import RefLib, RefUnavailable
    try:
        return RefLib()
    except RefUnavailable as e:
        if os.environ.get("MC_REQUIRE_REF") == "1":
            pytest.fail("MC_REQUIRE_REF=1 and the reference's kernels are not available: %s" % e)
        pytest.skip(str(e))
