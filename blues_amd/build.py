"""Builds the HIP engine in-tree: blues_amd/csrc/libblues_hip.so (gfx950 only)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("BLUES_LIB_PATH") or os.path.join(CSRC, "libblues_hip.so")   # (override: development builds with other compiler flags)
# -fno-slp-vectorize: the SLP pass packs pairs of fp32 operations into v_pk_* instructions, which issue at half rate on
# gfx950 (scripts/valu_issue.hip) and cost extra moves: the pair kernel runs 10 % faster without it (profiles/README.md)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-fno-slp-vectorize"]


def sources():
    """Everything the library is compiled from: the one translation unit and every header beside it."""
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))


def source_sha():
    """SHA-256 prefix of the sources AND the compiler flags: evidence collected from one build (PMC counters) is only
    quoted for the same build."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sources():
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def is_stale():
    if os.environ.get("BLUES_LIB_PATH"):
        return False
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in sources()] + [os.path.join(_HERE, "..", "include", "blues_engine.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_engine(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> libblues_hip.so; returns the library path."""
    if not force and not is_stale():
        return LIB_PATH
    # one builder at a time: with one process per GPU every rank may get here together
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():
                return LIB_PATH
            tmp = LIB_PATH + ".tmp.%d" % os.getpid()
            cmd = [hipcc_path()] + FLAGS + ["-o", tmp, os.path.join(CSRC, "blues_engine.hip")]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build_engine(force=True, verbose=True))
