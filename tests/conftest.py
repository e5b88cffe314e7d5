import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def emu():
    """Kernel sources compiled for the CPU emulator (tests/emu) — logic checks without a GPU."""
    from tests.emu import emu_native
    return emu_native.get()


@pytest.fixture(scope="session")
def gpu():
    """The product library on a real device.  torch is imported FIRST so that both share one HIP runtime."""
    import torch  # noqa: F401
    import tsxform
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    return tsxform.get()
