import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
