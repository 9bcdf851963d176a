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
def oracle():
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def pkg():
    import optix_test_smallpt_amd
    return optix_test_smallpt_amd


@pytest.fixture(scope="session")
def renderer(pkg):
    r = pkg.Renderer(0)
    r.set_watchdog(60.0)       # a scheduling bug in the persistent kernels must fail a test, not hang the GPU box
    yield r
    r.close()
