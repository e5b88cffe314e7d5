"""ctypes binding of the C ABI in include/tsxform.h (libtsxform.so, built in-tree by csrc/Makefile).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is usable, loading or
tsx_init() raises.  (The CPU emulator under tests/emu is a test harness for kernel logic and is never
loaded from here.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtsxform.so")

COMPRESS, ENCRYPT, CRC = 1, 2, 4
MEM_HOST, MEM_DEVICE, MEM_HOST_PACKED = 0, 1, 2
OK, E_INVAL, E_DEVICE, E_NOMEM, E_DST_TOO_SMALL, E_TAG_MISMATCH, E_BAD_FRAME, E_BAD_SIZE, E_SHORT_CHUNK, E_UNSUPPORTED = \
    0, -1, -2, -3, -4, -5, -6, -7, -8, -9
ZSTD_PROFILE_1_5_6, ZSTD_PROFILE_1_5_7 = 0, 1
ABI_VERSION = 4          # TSX_ABI_VERSION of include/tsxform.h these prototypes were written against


class ChunkDesc(C.Structure):
    _fields_ = [("src_off", C.c_uint64), ("dst_off", C.c_uint64), ("src_len", C.c_uint32), ("dst_cap", C.c_uint32),
                ("dst_len", C.c_uint32), ("crc32c", C.c_uint32), ("status", C.c_int32), ("iv", C.c_uint8 * 12)]


class BatchParams(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("aad_len", C.c_uint32), ("key", C.c_uint8 * 32), ("aad", C.c_uint8 * 64),
                ("zstd_level", C.c_int32), ("zstd_profile", C.c_uint32)]


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("h2d_ms", C.c_float), ("d2h_ms", C.c_float), ("crc_ms", C.c_float),
                ("zstd_ms", C.c_float), ("gcm_ms", C.c_float), ("unzstd_ms", C.c_float),
                ("crc_launches", C.c_uint32), ("zstd_launches", C.c_uint32), ("gcm_launches", C.c_uint32),
                ("unzstd_launches", C.c_uint32)]


class Config(C.Structure):
    """tsx_config of include/tsxform.h; fields left at CFG_DEFAULT take the library's default."""
    _fields_ = [("struct_size", C.c_uint32), ("fetch_reserved_cus", C.c_uint32), ("service_max_launch_ms", C.c_uint32), ("fetch_shared_cu_waves", C.c_uint32),
                ("pool_idle_bytes", C.c_uint64), ("fetch_quiet_ms", C.c_uint32), ("reserved2_", C.c_uint32)]


class ServiceInfo(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("watchdog_launches", C.c_uint64), ("members", C.c_uint64), ("chunks", C.c_uint64),
                ("kernel_ms", C.c_double), ("running", C.c_uint32), ("waves", C.c_uint32), ("compute_units", C.c_uint32),
                ("cu_keys_seen", C.c_uint32), ("reserved_cus", C.c_uint32), ("device_chunks", C.c_uint32), ("wave_starts", C.c_uint32),
                ("reserved_exits", C.c_uint32), ("skipped_tickets", C.c_uint32), ("live_waves", C.c_uint32), ("live_waves_max", C.c_uint32), ("shader_engines", C.c_uint32), ("rotations", C.c_uint32),
                ("guest_launches", C.c_uint32), ("yielded_waves", C.c_uint32), ("returned_chunks", C.c_uint32), ("readmissions", C.c_uint32), ("relocated_waves", C.c_uint32)]
assert C.sizeof(ServiceInfo) == 112


CFG_DEFAULT, CFG_DEFAULT64 = 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF
DESC_DTYPE = np.dtype([("src_off", "<u8"), ("dst_off", "<u8"), ("src_len", "<u4"), ("dst_cap", "<u4"), ("dst_len", "<u4"),
                       ("crc32c", "<u4"), ("status", "<i4"), ("iv", "u1", (12,))])
assert DESC_DTYPE.itemsize == C.sizeof(ChunkDesc) == 48

EXPORTS = ["tsx_abi_version", "tsx_version", "tsx_strerror", "tsx_init", "tsx_shutdown", "tsx_device_count",
           "tsx_ctx_create", "tsx_ctx_destroy", "tsx_ctx_timing", "tsx_transformed_bound", "tsx_transform_batch",
           "tsx_detransform_batch", "tsx_crc32c_batch", "tsx_device_malloc", "tsx_device_free", "tsx_memcpy_h2d",
           "tsx_memcpy_d2h", "tsx_ctx_device", "tsx_set_thread_device", "tsx_pool_stats", "tsx_host_register", "tsx_host_unregister",
           "tsx_init_ex", "tsx_service_stats", "tsx_service_quiesce"]


class TsxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d)" % (msg, code))
        self.code = code


class Native:
    """One loaded libtsxform with typed prototypes."""

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError("tsxform: HIP library %s is missing - run `make -C %s` (there is no CPU fallback)"
                               % (path, os.path.join(_HERE, "csrc")))
        L = C.CDLL(path)
        vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
        L.tsx_abi_version.restype = u32
        # ABI 3 put src_size in the MIDDLE of the batch entry points: a stale library called with these prototypes would take src_size
        # for the dst pointer.  Refuse it before any other symbol is touched (tests/test_boundary.py::test_stale_abi_is_refused).
        if not hasattr(L, "tsx_abi_version") or L.tsx_abi_version() != ABI_VERSION:
            raise RuntimeError("tsxform: %s speaks ABI %s, this binding ABI %d - rebuild it (`make -C %s`)"
                               % (path, L.tsx_abi_version() if hasattr(L, "tsx_abi_version") else "?", ABI_VERSION, os.path.join(_HERE, "csrc")))
        L.tsx_version.restype = C.c_char_p
        L.tsx_strerror.restype = C.c_char_p; L.tsx_strerror.argtypes = [C.c_int]
        L.tsx_init.restype = C.c_int; L.tsx_init.argtypes = [C.c_int, C.POINTER(C.c_int)]
        L.tsx_init_ex.restype = C.c_int; L.tsx_init_ex.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(Config)]
        L.tsx_service_stats.restype = C.c_int; L.tsx_service_stats.argtypes = [C.c_int, C.POINTER(ServiceInfo)]
        L.tsx_service_quiesce.restype = C.c_int; L.tsx_service_quiesce.argtypes = [C.c_int]
        L.tsx_debug_config.restype = C.c_longlong; L.tsx_debug_config.argtypes = [C.c_char_p, C.c_longlong]      # test hook, not in the header
        L.tsx_shutdown.restype = None
        L.tsx_device_count.restype = C.c_int
        L.tsx_ctx_create.restype = C.c_int; L.tsx_ctx_create.argtypes = [C.c_int, u32, u32, C.POINTER(vp)]
        L.tsx_ctx_destroy.restype = None; L.tsx_ctx_destroy.argtypes = [vp]
        L.tsx_ctx_timing.restype = C.c_int; L.tsx_ctx_timing.argtypes = [vp, C.POINTER(Timing)]
        L.tsx_transformed_bound.restype = sz; L.tsx_transformed_bound.argtypes = [sz, u32]
        for name in ("tsx_transform_batch", "tsx_detransform_batch"):
            f = getattr(L, name); f.restype = C.c_int
            f.argtypes = [vp, C.POINTER(BatchParams), vp, u32, vp, sz, vp, sz, C.c_int]
        L.tsx_crc32c_batch.restype = C.c_int; L.tsx_crc32c_batch.argtypes = [vp, vp, u32, vp, sz, C.c_int]
        L.tsx_device_malloc.restype = C.c_int; L.tsx_device_malloc.argtypes = [C.c_int, sz, C.POINTER(vp)]
        L.tsx_device_free.restype = C.c_int; L.tsx_device_free.argtypes = [C.c_int, vp]
        L.tsx_memcpy_h2d.restype = C.c_int; L.tsx_memcpy_h2d.argtypes = [C.c_int, vp, vp, sz]
        L.tsx_memcpy_d2h.restype = C.c_int; L.tsx_memcpy_d2h.argtypes = [C.c_int, vp, vp, sz]
        L.tsx_ctx_device.restype = C.c_int; L.tsx_ctx_device.argtypes = [vp]
        L.tsx_set_thread_device.restype = C.c_int; L.tsx_set_thread_device.argtypes = [C.c_int]
        L.tsx_pool_stats.restype = C.c_int
        L.tsx_pool_stats.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.tsx_host_register.restype = C.c_int; L.tsx_host_register.argtypes = [vp, sz]
        L.tsx_host_unregister.restype = C.c_int; L.tsx_host_unregister.argtypes = [vp]
        self.lib = L
        self.path = path
        self._inited = False

    # ---- lifetime -----------------------------------------------------------------------------
    def check(self, rc):
        if rc < 0:
            raise TsxError(rc, self.lib.tsx_strerror(rc).decode())
        return rc

    def init(self, device_count=0, device_ids=None, fetch_reserved_cus=None, service_max_launch_ms=None, pool_idle_bytes=None, fetch_shared_cu_waves=None, fetch_quiet_ms=None):
        ids = (C.c_int * len(device_ids))(*device_ids) if device_ids else None
        if fetch_reserved_cus is None and service_max_launch_ms is None and pool_idle_bytes is None and fetch_shared_cu_waves is None and fetch_quiet_ms is None:
            n = self.check(self.lib.tsx_init(device_count, ids))
        else:
            cfg = Config(C.sizeof(Config), CFG_DEFAULT if fetch_reserved_cus is None else fetch_reserved_cus,
                         CFG_DEFAULT if service_max_launch_ms is None else service_max_launch_ms, CFG_DEFAULT if fetch_shared_cu_waves is None else fetch_shared_cu_waves,
                         CFG_DEFAULT64 if pool_idle_bytes is None else pool_idle_bytes, CFG_DEFAULT if fetch_quiet_ms is None else fetch_quiet_ms, 0)
            n = self.check(self.lib.tsx_init_ex(device_count, ids, C.byref(cfg)))
        self._inited = True
        return n

    def service_stats(self, device_index=0):
        """Counters of the device's compressor service (tsx_service_info) as a dict."""
        s = ServiceInfo()
        self.check(self.lib.tsx_service_stats(device_index, C.byref(s)))
        return {k: getattr(s, k) for k, _ in ServiceInfo._fields_}

    def service_quiesce(self, device_index=0):
        self.check(self.lib.tsx_service_quiesce(device_index))

    def debug_config(self, key, value):
        """Test / measurement hook: set one configuration field of the loaded library, return the previous value."""
        old = self.lib.tsx_debug_config(key.encode(), int(value))
        if old == E_INVAL and key not in ("pool_idle_bytes",):
            raise KeyError(key)
        return old

    def configured(self, **kv):
        """Context manager: configuration fields set for the duration of a with-block (tests)."""
        nat = self

        class _Scope:
            def __enter__(self_):
                self_.old = {k: nat.debug_config(k, v) for k, v in kv.items()}
                return nat

            def __exit__(self_, *a):
                for k, v in self_.old.items():
                    nat.debug_config(k, v)
                return False
        return _Scope()

    def version(self):
        return self.lib.tsx_version().decode()

    def strerror(self, code):
        return self.lib.tsx_strerror(code).decode()

    def ctx_create(self, device_index=0, max_chunks=0, max_chunk_size=0):
        h = C.c_void_p()
        self.check(self.lib.tsx_ctx_create(device_index, max_chunks, max_chunk_size, C.byref(h)))
        return h

    def ctx_destroy(self, h):
        self.lib.tsx_ctx_destroy(h)

    def ctx_timing(self, h):
        t = Timing()
        self.check(self.lib.tsx_ctx_timing(h, C.byref(t)))
        return t

    def ctx_device(self, h):
        return self.check(self.lib.tsx_ctx_device(h))

    def set_thread_device(self, device_index):
        """Device of this thread's ctx-less calls (-1: least loaded)."""
        self.check(self.lib.tsx_set_thread_device(device_index))

    def pool_stats(self, device_index=0):
        idle, in_use, batches = C.c_uint32(), C.c_uint32(), C.c_uint64()
        self.check(self.lib.tsx_pool_stats(device_index, C.byref(idle), C.byref(in_use), C.byref(batches)))
        return {"idle": idle.value, "in_use": in_use.value, "batches": batches.value}

    def host_register(self, arr):
        self.check(self.lib.tsx_host_register(arr.ctypes.data, arr.nbytes))

    def host_unregister(self, arr):
        self.check(self.lib.tsx_host_unregister(arr.ctypes.data))

    def shutdown(self):
        self.lib.tsx_shutdown()
        self._inited = False

    def transformed_bound(self, n, flags):
        return self.lib.tsx_transformed_bound(n, flags)

    # ---- device memory ------------------------------------------------------------------------
    def device_malloc(self, nbytes, device_index=0):
        p = C.c_void_p()
        self.check(self.lib.tsx_device_malloc(device_index, nbytes, C.byref(p)))
        return p.value

    def device_free(self, p, device_index=0):
        self.check(self.lib.tsx_device_free(device_index, p))

    def h2d(self, dptr, arr, device_index=0):
        a = np.ascontiguousarray(arr)
        self.check(self.lib.tsx_memcpy_h2d(device_index, dptr, a.ctypes.data, a.nbytes))

    def d2h(self, arr, dptr, device_index=0):
        assert arr.flags["C_CONTIGUOUS"]
        self.check(self.lib.tsx_memcpy_d2h(device_index, arr.ctypes.data, dptr, arr.nbytes))

    # ---- batches ------------------------------------------------------------------------------
    @staticmethod
    def make_params(flags, key=b"", aad=b"", zstd_level=0, zstd_profile=ZSTD_PROFILE_1_5_7):
        p = BatchParams()
        p.flags = flags
        p.aad_len = len(aad)
        if len(aad) > 64:
            raise ValueError("aad longer than 64 bytes")
        if flags & ENCRYPT and len(key) != 32:
            raise ValueError("AES-256 key must be 32 bytes")
        C.memmove(p.key, bytes(key).ljust(32, b"\0"), 32)
        C.memmove(p.aad, bytes(aad).ljust(64, b"\0"), 64)
        p.zstd_level = zstd_level
        p.zstd_profile = zstd_profile
        return p

    @staticmethod
    def _device_ready(mem_kind):
        """TSX_MEM_DEVICE hands the library raw device pointers; its kernels run on the context's OWN streams and know nothing about
        the stream that produced those buffers.  A Python caller's buffers are usually torch tensors: whatever torch still has queued
        on its current stream (the kernel filling `src`, the memset of `dst`) must be complete first - include/tsxform.h says so."""
        if mem_kind == MEM_DEVICE:
            import sys
            torch = sys.modules.get("torch")
            if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
                torch.cuda.current_stream().synchronize()

    def _ptr(self, x):
        if x is None:
            return None
        if isinstance(x, np.ndarray):
            return x.ctypes.data
        return x  # raw device pointer (int)

    @staticmethod
    def _src_size(descs, src, src_size):
        """Size of the caller's source buffer (ABI 3).  A numpy array knows its own; for a raw device pointer the caller states it,
        or - src_size None - vouches that the descriptors lie inside the allocation (their extent is passed)."""
        if src_size is not None:
            return int(src_size)
        if isinstance(src, np.ndarray):
            return src.nbytes
        return int((descs["src_off"] + descs["src_len"]).max()) if len(descs) else 0

    def transform_batch(self, params, descs, src, dst, dst_size, mem_kind=MEM_HOST, ctx=None, src_size=None):
        assert descs.dtype == DESC_DTYPE and descs.flags["C_CONTIGUOUS"]
        self._device_ready(mem_kind)
        return self.check(self.lib.tsx_transform_batch(ctx, C.byref(params), descs.ctypes.data, len(descs), self._ptr(src),
                                                       self._src_size(descs, src, src_size), self._ptr(dst), dst_size, mem_kind))

    def detransform_batch(self, params, descs, src, dst, dst_size, mem_kind=MEM_HOST, ctx=None, src_size=None):
        assert descs.dtype == DESC_DTYPE and descs.flags["C_CONTIGUOUS"]
        self._device_ready(mem_kind)
        return self.check(self.lib.tsx_detransform_batch(ctx, C.byref(params), descs.ctypes.data, len(descs), self._ptr(src),
                                                         self._src_size(descs, src, src_size), self._ptr(dst), dst_size, mem_kind))

    def crc32c_batch(self, descs, src, mem_kind=MEM_HOST, ctx=None, src_size=None):
        assert descs.dtype == DESC_DTYPE and descs.flags["C_CONTIGUOUS"]
        self._device_ready(mem_kind)
        return self.check(self.lib.tsx_crc32c_batch(ctx, descs.ctypes.data, len(descs), self._ptr(src), self._src_size(descs, src, src_size), mem_kind))


_native = None


def get():
    """The process-wide product library (libtsxform.so, HIP).  Raises when it or the GPU is missing."""
    global _native
    if _native is None:
        n = Native(LIB_PATH)
        n.init()
        _native = n
    return _native
