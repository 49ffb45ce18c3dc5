import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "slam-module_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    import mso
    mso.build()
    return mso


@pytest.fixture(scope="session")
def ctx():
    """One device context for the whole GPU session; fails loudly when the HIP library / GPU is missing."""
    import mi355slam
    c = mi355slam.Context(0)
    yield c
    c.close()
