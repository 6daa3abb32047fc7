import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_bind
    return oracle_bind.Oracle()


@pytest.fixture(scope="session")
def engine():
    """One LcsGpu context for the GPU tests.  Fails (not skips) when the HIP library is missing."""
    import famsa_amd
    eng = famsa_amd.LcsGpu(0)
    yield eng
    eng.close()
