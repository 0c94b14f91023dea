import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()      # make: a no-op when liboracle.so is newer than its sources
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def hip_ctx():
    """A context on cuda:0. No fallback: if the library or the GPU is missing the test errors."""
    from ziren_amd import prover
    ctx = prover.Context(0)
    yield ctx
    ctx.close()
