import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine_lib():
    """The built HIP engine (built here if stale; hipcc cross-compiles without a GPU)."""
    from procgen2_amd import build as pgbuild
    from procgen2_amd import lib as pglib
    if not os.path.exists(pglib.DEFAULT_LIB):
        pgbuild.build(verbose=False)
    return pglib.load()


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_util
    return oracle_util.oracle()
