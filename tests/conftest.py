import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    try:
        from minivectordb_amd import _native
        return bool(_native.lib().mvdb_device_available())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: the product has no CPU fallback.
    pass


@pytest.fixture(scope="session")
def gpu():
    from minivectordb_amd import _native
    if not _native.lib().mvdb_device_available():
        pytest.fail("no HIP device visible: GPU tests cannot run (there is no CPU fallback)")
    return True
