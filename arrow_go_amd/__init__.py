"""arrow_go_amd — MI355X-native execution layer for arrow-go's compute/math hot path.

The product is `libarrowhip.so` (hand-written HIP for gfx950, C ABI in
include/arrowhip.h).  This package is the loader plus thin host-side handles used by
the tests, bench.py and the array-level mirror of arrow-go's `compute` API.  Importing
it fails loudly if the HIP library is missing: there is no CPU fallback.
"""
from . import _native  # noqa: F401  (raises ImportError if libarrowhip.so is absent)
from ._native import (ArrowHipError, ErrInvalid, ErrIndex, ErrOverflow, ErrHip, ErrNotImplemented)  # noqa: F401
from .device import Comm, Context, DeviceBuffer, Ingest, PinnedBuffer, device_count, expr_codegen  # noqa: F401

__version__ = "0.1.0"
