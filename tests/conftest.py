import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The test session binds the MEASUREMENT build of the library (libnemar_hip_ab.so: the product's sources with -DNEMAR_AB): the kernel
# and step tests lower route thresholds and switch routes through nemar_tune, which the product does not have.  The two builds share
# every default-path kernel byte for byte (tests/test_abi.py compares the code objects) and tests/test_product_lib_gpu.py runs the
# training step on the product library in a fresh process and compares it with this one bit for bit.  NEMAR_AB_LIBRARY=0 in the
# environment runs the session on the product library instead (the tests that need a switch then fail loudly).
os.environ.setdefault("NEMAR_AB_LIBRARY", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def emu_lib():
    """The kernel sources compiled for the host SIMT emulator (tests/emu) — CPU tier only."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from nemar_amd import _lib
    return _lib.load(build_emu.build())


@pytest.fixture(scope="session")
def hip_lib():
    from nemar_amd import _lib
    return _lib.load()
