import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def si():
    """The product package; builds libsubspace_hip.so in-tree when it is missing (hipcc cross-compiles)."""
    import subspaceinference_jl_amd as pkg
    from subspaceinference_jl_amd import build
    if not os.path.exists(pkg._capi.LIB_PATH):
        build.build()
    pkg.load()
    return pkg


@pytest.fixture(scope="session")
def gpu_ctx(si):
    ctx = si.Context(0)
    yield ctx
    ctx.close()
