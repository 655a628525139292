import os
import sys

import pytest


def _cpu_budget():
    """CPUs the container may really use (affinity capped by the cgroup quota: the GPU box shows 256 and grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


# the oracle is NumPy: keep its BLAS pool within the CPU budget (one spinning thread per VISIBLE CPU gets the whole
# process throttled by the cgroup for ~80 ms at a time)
os.environ.setdefault("OPENBLAS_NUM_THREADS", str(_cpu_budget()))

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
    dev_lib = os.environ.get("SI_TEST_LIB")   # the development build under test (guard-page runs: csrc/guard_alloc.hip)
    if dev_lib:
        pkg._capi.LIB_PATH = os.path.abspath(dev_lib)
    if not os.path.exists(pkg._capi.LIB_PATH):
        build.build()
    pkg.load()
    return pkg


@pytest.fixture(scope="session")
def gpu_ctx(si):
    ctx = si.Context(0)
    yield ctx
    ctx.close()
