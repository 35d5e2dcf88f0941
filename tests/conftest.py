import importlib
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    return importlib.import_module("rust-tracing_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib


def _has_gpu():
    try:
        rt = importlib.import_module("rust-tracing_amd")
        return rt.amd_lib().rt_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu(rt):
    """The GPU tests must run the HIP path: no GPU (or no library) is a failure, not a skip."""
    n = rt.amd_lib().rt_device_count()
    assert n > 0, "no HIP device visible: -m gpu tests need an MI355X"
    return n
