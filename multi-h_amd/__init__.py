"""multi-h_amd — MI355X-native engine for Multi-H's propose-score-label hot path.

The product is the C-ABI library `libmultih_hip.so` (include/multih_hip.h,
hand-written gfx950 HIP kernels in csrc/) and the C++ host class `MultiH`
(host/, mirroring the reference's M/MultiH.h).  This Python package is
plumbing only: a ctypes binding (capi.Engine), the synthetic scene generator
used by tests/bench, and the multi-GPU sharding helpers.

The directory name contains a hyphen (it is the name the project brief fixes),
so import it with importlib:

    import importlib; mh = importlib.import_module("multi-h_amd")
"""
from . import synth  # noqa: F401
from .capi import Engine, MultiHError, device_count, load_library, LIB_PATH, SYMBOLS  # noqa: F401

__all__ = ["Engine", "MultiHError", "device_count", "load_library", "synth", "LIB_PATH", "SYMBOLS"]
