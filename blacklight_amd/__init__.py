"""blacklight_amd: MI355X-native hot path of the blacklight general-relativistic ray tracer.

The product is the HIP library (blacklight_amd/libblacklight_amd.so, sources in csrc/) behind the
C-ABI in include/blacklight_amd.h; this package is the thin host layer around it.
"""
from . import mock  # noqa: F401
from ._capi import BlacklightError, LIB_PATH  # noqa: F401
from .context import Context  # noqa: F401
from .params import Params  # noqa: F401
from .snapshot import Snapshot  # noqa: F401

__all__ = ["Params", "Context", "Snapshot", "BlacklightError", "mock", "LIB_PATH"]
