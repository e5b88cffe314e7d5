"""TEST HARNESS: loads the CPU-emulated build of the kernel sources (tests/emu/_build/libtsxform_emu.so).
Only tests import this; the product package (`tsxform`) never does."""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
EMU_LIB = os.path.join(_HERE, "_build", "libtsxform_emu.so")
_emu = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "tiered-storage-for-apache-kafka_amd", "csrc"), "emu"],
                          stdout=subprocess.DEVNULL)
    return EMU_LIB


def get():
    global _emu
    if _emu is None:
        lib = os.environ.get("TSX_EMU_LIB")         # a private copy (long fuzz runs survive rebuilds of the in-tree file)
        if not lib:
            build()
            lib = EMU_LIB
        sys.path.insert(0, _ROOT)
        import tsxform
        n = tsxform._native.Native(lib)
        os.environ["TSX_ALLOW_ANY_ARCH"] = "1"      # the emulator reports arch "emu"
        n.init()
        _emu = n
    return _emu
