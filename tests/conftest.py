import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _prebuild_for_workers():
    """Everything the CPU tests compile on demand, once, before the workers start: their own `make` calls then find it up to date (several
    workers compiling one output at the same time would race).  A failure here is left for the test that needs the artefact to report."""
    import subprocess
    csrc = os.path.join(ROOT, "tiered-storage-for-apache-kafka_amd", "csrc")
    host = os.path.join(ROOT, "tests", "host")
    for cmd in (["make", "-s", "-C", csrc], ["make", "-s", "-C", csrc, "emu"], ["make", "-s", "-C", csrc, "emu-asan"], ["make", "-s", "-C", csrc, "emu-tsan"],
                ["make", "-s", "-C", os.path.join(ROOT, "oracle")], ["make", "-s", "-C", host], ["make", "-s", "-C", host, "tsan"]):
        try:
            subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1800)
        except Exception:                                  # noqa: BLE001
            pass


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`, the driver's command) is 22 minutes of emulated kernels in ONE process: it runs in four pytest-xdist
    workers instead, one test FILE at a time per worker (a file's tests keep their order and never run side by side), ~7 minutes.
    Only that selection: the `-m gpu` suite has one device and stays one process.  An explicit `-n`, `TSX_TEST_WORKERS=0`, `--pdb`,
    `--collect-only` or a missing pytest-xdist leave everything as it was."""
    opt = config.option
    if os.environ.get("PYTEST_XDIST_WORKER") or not config.pluginmanager.hasplugin("xdist"):
        return None
    if (getattr(opt, "markexpr", "") or "").strip() != "not gpu" or getattr(opt, "numprocesses", None) is not None:
        return None
    if getattr(opt, "usepdb", False) or getattr(opt, "collectonly", False):
        return None
    try:
        workers = int(os.environ.get("TSX_TEST_WORKERS", "4"))
    except ValueError:
        workers = 4
    workers = min(workers, max(1, (os.cpu_count() or 2) // 2))
    if workers < 2:
        return None
    _prebuild_for_workers()
    opt.numprocesses = workers
    opt.dist = "loadfile"
    return None


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
