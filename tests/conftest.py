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
    """The CPU oracle (test infrastructure; oracle/kaamer_oracle.c)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def klib():
    """The product library, built in-tree if stale."""
    if not os.environ.get("KAAMER_HOST_ONLY"):   # (tools/asan runs the host tests against its own sanitized library)
        from kaamer_amd import build
        build.build()
    from kaamer_amd import abi
    return abi.lib()


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no GPU is visible")
    return 0
