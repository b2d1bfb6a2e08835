"""ctypes binding of the CPU oracle (oracle/blues_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from blues_amd/.  Parity status: see the
header of blues_oracle.h ("parity unpinned" for forces/energies/work).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from blues_amd._abi import (BluesIntegratorDesc, BluesSystemDesc, IntegratorData, SystemData, N_ENERGY_TERMS)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libblues_oracle.so")
_LIB_PATH_OMP = os.path.join(_HERE, "_build", "libblues_oracle_omp.so")   # pair loop on all cores (cpu_baseline only)
_dp = C.POINTER(C.c_double)
_lib = None
_lib_omp = None


def build(force=False):
    src = os.path.join(_HERE, "blues_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or not os.path.exists(_LIB_PATH_OMP) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def usable_cores():
    """CPU cores this process can actually keep busy: its affinity mask, capped by the cgroup CPU quota (a container that SEES 256
    cores may be granted 16 cores' worth of time; 192 OpenMP threads then run slower than 16)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, -(-int(txt[0]) // int(txt[1]))))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, -(-q // per)))
            break
        except Exception:
            continue
    return n


def lib(openmp=False):
    global _lib, _lib_omp
    if openmp:
        if _lib_omp is None:
            if not os.path.exists(_LIB_PATH_OMP):
                build()
            _lib_omp = _declare(C.CDLL(_LIB_PATH_OMP))
            _lib_omp.orc_set_threads.argtypes = [C.c_int]; _lib_omp.orc_set_threads.restype = None
            _lib_omp.orc_set_threads(int(os.environ.get("OMP_NUM_THREADS", 0)) or usable_cores())
        return _lib_omp
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _declare(C.CDLL(_LIB_PATH))
    return _lib


def _declare(L):
    if True:   # (argument and result types of every entry point)
        H = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(BluesSystemDesc), C.POINTER(BluesIntegratorDesc)]; L.orc_create.restype = H
        L.orc_destroy.argtypes = [H]; L.orc_destroy.restype = None
        L.orc_last_error.restype = C.c_char_p
        for name in ("orc_set_positions", "orc_set_velocities", "orc_get_positions", "orc_get_velocities", "orc_set_box", "orc_get_forces"):
            getattr(L, name).argtypes = [H, _dp]; getattr(L, name).restype = None
        L.orc_set_velocities_to_temperature.argtypes = [H, C.c_double, C.c_uint64]; L.orc_set_velocities_to_temperature.restype = None
        L.orc_energy_forces.argtypes = [H, C.c_double, C.c_double, _dp, _dp]; L.orc_energy_forces.restype = C.c_double
        L.orc_potential_energy.argtypes = [H]; L.orc_potential_energy.restype = C.c_double
        L.orc_kinetic_energy.argtypes = [H]; L.orc_kinetic_energy.restype = C.c_double
        L.orc_step.argtypes = [H, C.c_int]; L.orc_step.restype = C.c_int
        L.orc_get_global.argtypes = [H, C.c_char_p, _dp]; L.orc_get_global.restype = C.c_int
        L.orc_set_global.argtypes = [H, C.c_char_p, C.c_double]; L.orc_set_global.restype = C.c_int
        L.orc_reset.argtypes = [H]; L.orc_reset.restype = None
        L.orc_num_evaluations.argtypes = [H]; L.orc_num_evaluations.restype = C.c_long
        L.orc_set_custom_pair_mode.argtypes = [H, C.c_int]; L.orc_set_custom_pair_mode.restype = None
        L.orc_add_centroid_bond.argtypes = [H, C.c_int, C.POINTER(C.c_int), _dp, C.c_int, C.POINTER(C.c_int), _dp, C.c_double]; L.orc_add_centroid_bond.restype = C.c_int
        L.orc_minimize.argtypes = [H, C.c_int, C.c_double]; L.orc_minimize.restype = C.c_double
        L.orc_ewald_reciprocal_exact.argtypes = [H, C.c_int]; L.orc_ewald_reciprocal_exact.restype = C.c_double
        L.orc_default_lambda_sterics.argtypes = [C.c_double]; L.orc_default_lambda_sterics.restype = C.c_double
        L.orc_default_lambda_electrostatics.argtypes = [C.c_double]; L.orc_default_lambda_electrostatics.restype = C.c_double
        L.orc_get_prop_lambda.argtypes = [C.c_double, _dp]; L.orc_get_prop_lambda.restype = None
        L.orc_calculate_ncmc_steps.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]; L.orc_calculate_ncmc_steps.restype = None
        L.orc_philox4x32.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]; L.orc_philox4x32.restype = None
        L.orc_gaussians.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _dp]; L.orc_gaussians.restype = None
        L.orc_pair_energy.argtypes = [C.c_double] * 5 + [C.c_int] + [C.c_double] * 3 + [_dp]; L.orc_pair_energy.restype = C.c_double
    return L


def _ptr(a):
    return a.ctypes.data_as(_dp)


class Oracle:
    """One alchemical NCMC context + integrator on the CPU (fp64)."""

    def __init__(self, system: SystemData, integrator: IntegratorData, openmp=False):
        self._L = lib(openmp)
        sd, self._keep_s = system.to_desc()
        idesc, self._keep_i = integrator.to_desc()
        self._h = self._L.orc_create(C.byref(sd), C.byref(idesc))
        if not self._h:
            raise RuntimeError("oracle: " + self._L.orc_last_error().decode())
        self.n = system.n_atoms
        if system.positions is not None:
            self.set_positions(system.positions)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_destroy(self._h); self._h = None

    def set_positions(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.n, 3); self._L.orc_set_positions(self._h, _ptr(x))

    def set_velocities(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(self.n, 3); self._L.orc_set_velocities(self._h, _ptr(v))

    def get_positions(self):
        x = np.empty((self.n, 3)); self._L.orc_get_positions(self._h, _ptr(x)); return x

    def get_velocities(self):
        v = np.empty((self.n, 3)); self._L.orc_get_velocities(self._h, _ptr(v)); return v

    def set_box(self, box3):
        b = np.zeros(9); b[0], b[4], b[8] = box3; self._L.orc_set_box(self._h, _ptr(b))

    def set_velocities_to_temperature(self, T, seed):
        self._L.orc_set_velocities_to_temperature(self._h, float(T), int(seed))

    def energy_forces(self, lambda_sterics=1.0, lambda_electrostatics=1.0, forces=True):
        f = np.zeros((self.n, 3)) if forces else None
        t = np.zeros(N_ENERGY_TERMS)
        e = self._L.orc_energy_forces(self._h, float(lambda_sterics), float(lambda_electrostatics),
                                      _ptr(f) if forces else None, _ptr(t))
        return e, f, t

    def potential_energy(self):
        return self._L.orc_potential_energy(self._h)

    def kinetic_energy(self):
        return self._L.orc_kinetic_energy(self._h)

    def get_forces(self):
        f = np.empty((self.n, 3)); self._L.orc_get_forces(self._h, _ptr(f)); return f

    def step(self, n=1):
        rc = self._L.orc_step(self._h, int(n))
        if rc:
            raise RuntimeError("Particle coordinate is nan")

    def get_global(self, name):
        v = C.c_double()
        if self._L.orc_get_global(self._h, name.encode(), C.byref(v)):
            raise KeyError(name)
        return v.value

    def set_global(self, name, value):
        if self._L.orc_set_global(self._h, name.encode(), float(value)):
            raise KeyError(name)

    def reset(self):
        self._L.orc_reset(self._h)

    def num_evaluations(self):
        return self._L.orc_num_evaluations(self._h)

    def set_custom_pair_mode(self, mode):
        self._L.orc_set_custom_pair_mode(self._h, int(mode))

    def add_centroid_bond(self, idx1, w1, idx2, w2, k):
        i1 = (C.c_int * len(idx1))(*idx1); i2 = (C.c_int * len(idx2))(*idx2)
        a1 = np.ascontiguousarray(w1, dtype=np.float64); a2 = np.ascontiguousarray(w2, dtype=np.float64)
        if self._L.orc_add_centroid_bond(self._h, len(idx1), i1, _ptr(a1), len(idx2), i2, _ptr(a2), float(k)):
            raise RuntimeError("too many centroid bonds")

    def ewald_reciprocal_exact(self, mmax=12):
        return self._L.orc_ewald_reciprocal_exact(self._h, int(mmax))

    def minimize(self, max_iter=200, step0=0.01):
        return self._L.orc_minimize(self._h, int(max_iter), float(step0))


def philox4x32(c, k):
    out = (C.c_uint32 * 4)()
    lib().orc_philox4x32(c[0], c[1], c[2], c[3], k[0], k[1], out)
    return [int(v) for v in out]


def gaussians(seed, stream, draw, atom):
    out = np.zeros(3)
    lib().orc_gaussians(int(seed), int(stream), int(draw), int(atom), _ptr(out))
    return out


def pair_energy(r, qq, sigma, epsilon, alpha_ewald=0.0, alchemical=False, lambda_s=1.0, lambda_e=1.0, softcore_alpha=0.5):
    d = C.c_double()
    e = lib().orc_pair_energy(r, qq, sigma, epsilon, alpha_ewald, int(alchemical), lambda_s, lambda_e, softcore_alpha, C.byref(d))
    return e, d.value


def default_lambda_sterics(l):
    return lib().orc_default_lambda_sterics(float(l))


def default_lambda_electrostatics(l):
    return lib().orc_default_lambda_electrostatics(float(l))


def get_prop_lambda(p):
    out = np.zeros(2); lib().orc_get_prop_lambda(float(p), _ptr(out)); return float(out[0]), float(out[1])


def calculate_ncmc_steps(nstepsNC, nprop=1, prop_lambda=0.3):
    out = (C.c_int * 3)()
    lib().orc_calculate_ncmc_steps(int(nstepsNC), int(nprop), float(prop_lambda), out)
    return int(out[0]), int(out[1]), int(out[2])
