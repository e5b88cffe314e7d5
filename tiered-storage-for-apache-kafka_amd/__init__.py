"""tsxform — MI355X-native chunk-transform path for Kafka tiered storage.

Host-side mirror of the reference's transform/detransform interfaces over the C ABI of libtsxform.so.
"""
from . import _native  # noqa: F401
from ._native import (COMPRESS, CRC, ENCRYPT, MEM_DEVICE, MEM_HOST, TsxError, get)  # noqa: F401
HAVE_ZSTD = True  # Zstd level-3 compressor + frame decoder kernels are built into libtsxform
