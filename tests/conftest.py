import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The real reference behind oracle/ref_harness.cpp; only where oracle/Makefile built it."""
    from oracle_lib import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libpcramp_ref.so not built (no /root/reference here)")
    return Reference()
