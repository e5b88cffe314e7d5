#!/usr/bin/env python3
"""The same question as copy_engine.hip inside a python process that has torch loaded (bench.py, tools/broker_probe.py and the GPU tests
are such processes; a broker's JVM is not): 256 device -> host copies of 1.3 MB through hipMemcpyAsync of the runtime torch loaded, into a
torch-pinned tensor and into a numpy array registered with tsx_host_register.  Run under rocprofv3 --kernel-trace --memory-copy-trace."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "registered"
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
dev = torch.device("cuda", 0)
n, sz = 256, 1300000
src = torch.ones(64 * sz, dtype=torch.uint8, device=dev)
if mode == "torchpinned":
    dst_t = torch.empty(64 * sz, dtype=torch.uint8).pin_memory(); dptr = dst_t.data_ptr()
else:
    import tsxform
    N = tsxform._native.Native(); N.init(1, [0])
    dst = np.zeros(64 * sz, np.uint8); N.host_register(dst); dptr = dst.ctypes.data
torch.cuda.synchronize()
st = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0
for w in range(2):
    t0 = time.perf_counter()
    for i in range(n):
        off = (i % 64) * sz
        assert hip.hipMemcpyAsync(dptr + off, src.data_ptr() + off, sz, 2, st) == 0
    assert hip.hipStreamSynchronize(st) == 0
    el = time.perf_counter() - t0
print("torch process, d2h %s: %d x %d B in %.3f ms = %.2f GB/s" % (mode, n, sz, el * 1e3, n * sz / el / 1e9))
