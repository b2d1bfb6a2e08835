"""Loads the HIP engine (libblues_hip.so).  There is no CPU fallback: if the library is
missing or no HIP device is present, engine creation raises."""
import ctypes as C
import os

from . import _abi
from .build import LIB_PATH

_lib = None


class EngineUnavailable(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineUnavailable(
                "blues_amd: %s not found -- build it with `python -m blues_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        _abi.declare_engine_prototypes(lib)
        if lib.blues_abi_version() != _abi.ABI_VERSION:
            raise EngineUnavailable("blues_amd: ABI version mismatch")
        _lib = lib
        from . import tuning
        tuning.apply_environment()   # BLUES_TUNING="field=value,..." (the native library itself reads no environment)
    return _lib
