"""Launch-policy overrides of the engine (include/blues_engine.h: BluesTuning).

The engine chooses its own decomposition; tests and profiling runs pin one with

    with tuning.override(skin=0.12, k2_jiter=4, fork=0):
        engines = [...]          # engines and batches created inside keep the tuning for their lifetime

The native library reads no environment variables.  For command-line runs of scripts and bench.py the Python loader honours
ONE variable, ``BLUES_TUNING="fork=0,skin=0.12"`` (the same field names), applied once when the library is loaded.
"""
import contextlib
import os

from . import _abi, _lib

FIELDS = tuple(name for name, _ in _abi.BluesTuning._fields_ if name != "struct_size")


def defaults():
    t = _abi.BluesTuning()
    _lib.load().blues_tuning_default(t)
    return t


def current():
    t = _abi.BluesTuning()
    _lib.load().blues_get_tuning(t)
    return t


def _apply(t):
    if _lib.load().blues_set_tuning(t) != 0:
        raise RuntimeError("blues_set_tuning failed: %s" % (_lib.load().blues_last_error(None) or b"").decode())


def set(**fields):
    """Changes fields of the process-wide tuning (engines and batches created afterwards)."""
    t = current()
    for k, v in fields.items():
        if k not in FIELDS:
            raise KeyError("BluesTuning has no field %r (fields: %s)" % (k, ", ".join(FIELDS)))
        setattr(t, k, v)
    _apply(t)


def reset():
    _lib.load().blues_set_tuning(None)


@contextlib.contextmanager
def override(**fields):
    saved = current()
    set(**fields)
    try:
        yield
    finally:
        _apply(saved)


def as_dict():
    t = current()
    return {k: getattr(t, k) for k in FIELDS}


def parse(spec):
    """"fork=0,skin=0.12" -> {"fork": 0, "skin": 0.12}"""
    out = {}
    types = dict(_abi.BluesTuning._fields_)
    for item in filter(None, (s.strip() for s in spec.split(","))):
        k, _, v = item.partition("=")
        k = k.strip()
        if k not in FIELDS:
            raise KeyError("BLUES_TUNING: unknown field %r (fields: %s)" % (k, ", ".join(FIELDS)))
        import ctypes as C
        out[k] = float(v) if types[k] is C.c_double else int(v)
    return out


def apply_environment():
    spec = os.environ.get("BLUES_TUNING")
    if spec:
        set(**parse(spec))
