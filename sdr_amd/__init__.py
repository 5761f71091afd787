"""sdr_amd -- MI355X (gfx950) implementation of the erithion/SDR receive-side hot path.

The compute lives in libsdr_amd.so (hand-written HIP, C ABI in include/sdr_amd.h); this package
only binds it for Python callers (tests, bench, the BER driver).  No CPU fallback exists.
"""
from . import capi  # noqa: F401
from .capi import SdrError  # noqa: F401

__all__ = ["capi", "SdrError", "Context", "link"]


def __getattr__(name):
    if name == "Context":
        from .ops import Context
        return Context
    if name == "link":
        import importlib
        return importlib.import_module(".link", __name__)
    raise AttributeError(name)
