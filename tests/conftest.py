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
    # torch (used as plumbing by the resident-pipeline tests) must bring its HIP runtime up before
    # libmvdb_hip.so touches the device; the other order leaves torch with "No HIP GPUs are available".
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    from minivectordb_amd import _native
    if not _native.lib().mvdb_device_available():
        pytest.fail("no HIP device visible: GPU tests cannot run (there is no CPU fallback)")
    return True
