"""Thin ctypes wrapper over the C-ABI of include/blues_engine.h (one handle per replica)."""
import ctypes as C

import numpy as np

from . import _abi
from ._lib import load

_dp = C.POINTER(C.c_double)


class EngineError(RuntimeError):
    """Raised for any nonzero return of the C-ABI; carries blues_last_error()."""


class NativeEngine:
    def __init__(self, system: _abi.SystemData, integrator: _abi.IntegratorData, device=0):
        self._lib = load()
        sd, self._keep_s = system.to_desc()
        idesc, self._keep_i = integrator.to_desc()
        h = C.c_void_p()
        rc = self._lib.blues_engine_create(C.byref(sd), C.byref(idesc), int(device), C.byref(h))
        if rc:
            raise EngineError(self._lib.blues_last_error(None).decode())
        self._h = h
        self.n = system.n_atoms
        self.system = system
        self.integrator = integrator
        if system.positions is not None:
            self.set_positions(system.positions)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.blues_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise EngineError(self._lib.blues_last_error(self._h).decode())

    @staticmethod
    def _ptr(a):
        return a.ctypes.data_as(_dp)

    def set_positions(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.n, 3)
        self._check(self._lib.blues_set_positions(self._h, self._ptr(x), self.n))

    def set_velocities(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(self.n, 3)
        self._check(self._lib.blues_set_velocities(self._h, self._ptr(v), self.n))

    def snapshot(self, positions=True, velocities=True):
        return DeviceSnapshot(self, (1 if positions else 0) | (2 if velocities else 0))

    def set_positions_from_snapshot(self, snap):
        self._check(self._lib.blues_set_positions_from_snapshot(self._h, snap._h))

    def set_positions_from_snapshot_edited(self, snap, indices, xyz):
        """The snapshot's positions with a few atoms replaced.  False: not applicable (an edited atom is constrained to an
        unedited one) -- nothing was done, use set_positions."""
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        x = np.ascontiguousarray(xyz, dtype=np.float64).reshape(len(idx), 3)
        rc = self._lib.blues_set_positions_from_snapshot_edited(self._h, snap._h, idx.ctypes.data_as(C.POINTER(C.c_int32)), len(idx), x.ctypes.data_as(_dp))
        if rc == 3:
            return False
        self._check(rc)
        return True

    def set_velocities_from_snapshot(self, snap):
        self._check(self._lib.blues_set_velocities_from_snapshot(self._h, snap._h))

    def set_box(self, box3):
        # a full 3x3 matrix goes through as it is: the C side rejects non-orthorhombic boxes (include/blues_engine.h)
        self._box_key = None   # (Context.setPeriodicBoxVectors remembers what it set last; anybody else's box invalidates that)
        self.__dict__["_box_copy"] = None
        b = np.zeros(9)
        if np.size(box3) == 3:
            b[0], b[4], b[8] = np.asarray(box3, dtype=np.float64).reshape(-1)
        else:
            b[:] = np.asarray(box3, dtype=np.float64).reshape(9)
        self._check(self._lib.blues_set_box(self._h, self._ptr(b)))

    def get_positions(self):
        x = np.empty((self.n, 3)); self._check(self._lib.blues_get_positions(self._h, self._ptr(x), self.n)); return x

    def get_velocities(self):
        v = np.empty((self.n, 3)); self._check(self._lib.blues_get_velocities(self._h, self._ptr(v), self.n)); return v

    def get_forces(self):
        f = np.empty((self.n, 3)); self._check(self._lib.blues_get_forces(self._h, self._ptr(f), self.n)); return f

    def get_box(self):
        # (asked for by every getState; it changes only through set_box, which drops the copy kept here)
        b = self.__dict__.get("_box_copy")
        if b is None:
            b = np.zeros(9); self._check(self._lib.blues_get_box(self._h, self._ptr(b))); b = b.reshape(3, 3)
            self.__dict__["_box_copy"] = b
        return b.copy()

    def set_velocities_to_temperature(self, T, seed=0):
        self._check(self._lib.blues_set_velocities_to_temperature(self._h, float(T), int(seed) & 0xFFFFFFFFFFFFFFFF))

    def energies(self):
        """(potential, kinetic) in one call (state.getPotentialEnergy / getKineticEnergy of one getState)."""
        e = C.c_double(); k = C.c_double(); self._check(self._lib.blues_get_energy(self._h, C.byref(e), C.byref(k))); return e.value, k.value

    def potential_energy_at(self, lambda_sterics, lambda_electrostatics):
        """Potential energy of the current coordinates at other alchemical parameters; the engine's own stay as they are."""
        e = C.c_double(); self._check(self._lib.blues_get_energy_at(self._h, float(lambda_sterics), float(lambda_electrostatics), C.byref(e))); return e.value

    def potential_energy(self):
        e = C.c_double(); self._check(self._lib.blues_get_energy(self._h, C.byref(e), None)); return e.value

    def kinetic_energy(self):
        k = C.c_double(); self._check(self._lib.blues_get_energy(self._h, None, C.byref(k))); return k.value

    def energy_terms(self):
        t = np.zeros(_abi.N_ENERGY_TERMS); self._check(self._lib.blues_get_energy_terms(self._h, self._ptr(t))); return t

    def mesh_energy(self, with_alchemical_charges=False):
        """Reciprocal-space mesh energy of the current coordinates (kJ/mol; 0 without PME): include/blues_engine.h, blues_mesh_energy."""
        e = C.c_double(); self._check(self._lib.blues_mesh_energy(self._h, 1 if with_alchemical_charges else 0, C.byref(e))); return e.value

    def step(self, n=1):
        self.__dict__["_gcache"] = {}
        self._check(self._lib.blues_step(self._h, int(n)))

    def run_switch(self, n, trace=False):
        self.__dict__["_gcache"] = {}
        if trace:
            w = np.zeros(int(n)); self._check(self._lib.blues_run_switch(self._h, int(n), self._ptr(w))); return w
        self._check(self._lib.blues_run_switch(self._h, int(n), None))
        return None

    # The alchemical parameters are host-side mirrors in the engine and change only when the integrator steps or somebody sets
    # them; BLUES asks for them six times per chain and iteration (getState(getParameters=True), simulation.py:874-881), so the
    # last values are kept here and dropped by everything that can change them (step / run_switch / set_global / reset, and the
    # batch's step).
    _CACHED_GLOBALS = ("lambda_sterics", "lambda_electrostatics")

    def get_global(self, name):
        cache = self.__dict__.setdefault("_gcache", {})
        if name in cache:
            return cache[name]
        v = C.c_double(); self._check(self._lib.blues_get_global(self._h, name.encode(), C.byref(v)))
        if name in self._CACHED_GLOBALS:
            cache[name] = v.value
        return v.value

    def set_global(self, name, value):
        self.__dict__["_gcache"] = {}
        self._check(self._lib.blues_set_global(self._h, name.encode(), float(value)))

    def reset(self):
        self.__dict__["_gcache"] = {}
        self._check(self._lib.blues_reset(self._h))

    def stats(self):
        s = (C.c_int64 * _abi.N_STATS)(); self._check(self._lib.blues_get_stats(self._h, s))
        return {"force_passes": s[0], "list_generation": s[1], "kernel_launches": s[2], "i_tiles": s[3],
                "clusters": s[4], "jcap": s[5], "npart": s[6], "seg_len": s[7] // 1000, "wpb": s[7] % 1000,
                "max_jcount": s[8], "resorts": s[9], "list_builds": s[10], "own_energy_evaluations": s[11],
                "nonbonded_kernel": s[12], "tiles_per_list": s[13], "atom_list_entries": s[14], "atom_list_iterations": s[15],
                "atom_prunes": s[16], "pruned_list_entries": s[17], "pruned_list_iterations": s[18], "pruned_lists": s[19], "alchemical_kernel": s[20], "step_threads": s[21]}

    def time_nonbonded(self, reps=20):
        u = C.c_double(); self._check(self._lib.blues_time_nonbonded(self._h, int(reps), C.byref(u))); return u.value

    def time_list_build(self, reps=20):
        u = C.c_double(); self._check(self._lib.blues_time_list_build(self._h, int(reps), C.byref(u))); return u.value

    def audit_lists(self):
        """(pairs within the cutoff, pairs missing from the lists the nonbonded kernel would walk now) -- include/blues_engine.h: blues_audit_lists."""
        out = (C.c_int64 * 2)(); self._check(self._lib.blues_audit_lists(self._h, out)); return int(out[0]), int(out[1])


class DeviceSnapshot:
    """openmm.State's positions / velocities kept on the GPU (include/blues_engine.h "Device-resident State")."""

    def __init__(self, engine, what, handle=None):
        """handle: a snapshot already captured for this engine (NativeBatch.snapshot_all)."""
        self.engine, self.what = engine, int(what)
        if handle is None:
            handle = C.c_void_p()
            engine._check(engine._lib.blues_snapshot_capture(engine._h, self.what, C.byref(handle)))
        self._h = handle
        self._host = {}

    def read(self, kind):
        """kind 1: positions, 2: velocities -> (n, 3) float64 (downloaded once)."""
        if kind not in self._host:
            out = np.empty((self.engine.n, 3))
            self.engine._check(self.engine._lib.blues_snapshot_read(self._h, kind, out.ctypes.data_as(_dp), self.engine.n))
            self._host[kind] = out
        return self._host[kind]

    def read_atoms(self, kind, indices):
        """Coordinates (kind 1) or velocities (2) of a few atoms -> (len(indices), 3)."""
        if kind in self._host:
            return self._host[kind][np.asarray(indices, dtype=np.int64)].copy()
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        out = np.empty((len(idx), 3))
        self.engine._check(self.engine._lib.blues_snapshot_read_atoms(self._h, kind, idx.ctypes.data_as(C.POINTER(C.c_int32)), len(idx), out.ctypes.data_as(_dp)))
        return out

    def release(self):
        if getattr(self, "_h", None) and getattr(self.engine, "_h", None):
            self.engine._lib.blues_snapshot_release(self._h)
        self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class NativeBatch:
    """Replica batch (include/blues_engine.h, "Replica batches"): congruent engines on one GPU whose integrator steps
    share every kernel launch.  Each engine keeps its own interface; only stepping goes through here."""

    def __init__(self, engines):
        self._lib = load()
        self.engines = list(engines)
        arr = (C.c_void_p * len(self.engines))(*[e._h for e in self.engines])
        h = C.c_void_p()
        rc = self._lib.blues_batch_create(arr, len(self.engines), C.byref(h))
        if rc:
            raise EngineError(self._lib.blues_batch_last_error(None).decode())
        self._h = h
        # several batches on one GPU, each driven from its own host thread (simulation.BatchedBLUESSimulation(device_turn=...)): the
        # batches take TURNS on the device for their stepping calls -- one batch's kernels have the GPU to themselves while the
        # other batches' threads do their per-chain host work (hooks, state tables, Metropolis tests)
        self.device_turn = None

    def __len__(self):
        return len(self.engines)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.blues_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, n=1, trace=False, raise_errors=True, active=None):
        """integrator.step(n) on every member (or on those with active[r] true; the others are left untouched).  Returns
        (errors, traces): errors[r] is None or the EngineError member r raised; with raise_errors the first one is
        raised instead (single-replica behaviour)."""
        R = len(self.engines)
        mask = (C.c_int32 * R)(*[1 if (active is None or active[r]) else 0 for r in range(R)])
        if self._lib.blues_batch_set_active(self._h, mask):
            raise EngineError(self._lib.blues_batch_last_error(self._h).decode())
        status = (C.c_int32 * R)()
        w = np.zeros((R, int(n))) if trace else None
        for e in self.engines:
            e.__dict__["_gcache"] = {}
        if self.device_turn is not None:
            with self.device_turn:      # (the call blocks until the steps are done and releases the GIL meanwhile)
                rc = self._lib.blues_batch_step(self._h, int(n), w.ctypes.data_as(_dp) if trace else None, status)
        else:
            rc = self._lib.blues_batch_step(self._h, int(n), w.ctypes.data_as(_dp) if trace else None, status)
        if rc:
            raise EngineError(self._lib.blues_batch_last_error(self._h).decode())
        errors = [EngineError(self._lib.blues_last_error(e._h).decode()) if status[r] else None for r, e in enumerate(self.engines)]
        if raise_errors:
            for err in errors:
                if err is not None:
                    raise err
        return errors, w

    def prefetch_energies(self, potential=True, kinetic=True, active=None, at_lambda_one=False):
        """Evaluate the members' energies with shared launches into their caches (see include/blues_engine.h)."""
        R = len(self.engines)
        mask = (C.c_int32 * R)(*[1 if (active is None or active[r]) else 0 for r in range(R)])
        what = (1 if potential else 0) | (2 if kinetic else 0) | (4 if at_lambda_one else 0)
        if self._lib.blues_batch_set_active(self._h, mask) or self._lib.blues_batch_prefetch_energies(self._h, what):
            raise EngineError(self._lib.blues_batch_last_error(self._h).decode())

    # ---- the plugin boundary for all members at once (include/blues_engine.h, "The plugin boundary for a whole batch")
    def _mask(self, active):
        R = len(self.engines)
        return None if active is None else (C.c_int32 * R)(*[1 if active[r] else 0 for r in range(R)])

    def _fail(self):
        raise EngineError(self._lib.blues_batch_last_error(self._h).decode())

    def snapshot_all(self, positions=True, velocities=True, active=None):
        """[DeviceSnapshot or None per member]: every member's positions / velocities captured with one launch."""
        R = len(self.engines)
        what = (1 if positions else 0) | (2 if velocities else 0)
        out = (C.c_void_p * R)()
        if self._lib.blues_batch_snapshot_capture(self._h, what, self._mask(active), out):
            self._fail()
        return [DeviceSnapshot(e, what, handle=C.c_void_p(out[r])) if out[r] else None for r, e in enumerate(self.engines)]

    def _handles(self, snaps):
        R = len(self.engines)
        return (C.c_void_p * R)(*[(s._h if s is not None else None) for s in snaps])

    def restore_all(self, snaps, positions=True, velocities=True):
        """setPositions / setVelocities of every member from its snapshot (None: the member sits out).  False: some member cannot
        take the device route -- nothing was done, use the member calls."""
        for e, s in zip(self.engines, snaps):
            if s is not None:
                e.__dict__["_gcache"] = {}
        rc = self._lib.blues_batch_restore(self._h, self._handles(snaps), (1 if positions else 0) | (2 if velocities else 0))
        if rc == 3:
            return False
        if rc:
            self._fail()
        return True

    def restore_edited_all(self, snaps, indices, xyz):
        """setPositions of every member from its snapshot with the atoms `indices` overwritten by xyz[r] (nm)."""
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(len(self.engines), len(idx), 3)
        for e, s in zip(self.engines, snaps):
            if s is not None:
                e.__dict__["_gcache"] = {}
        rc = self._lib.blues_batch_restore_edited(self._h, self._handles(snaps), idx.ctypes.data_as(C.POINTER(C.c_int32)), len(idx), xyz.ctypes.data_as(_dp))
        if rc == 3:
            return False
        if rc:
            self._fail()
        return True

    def read_atoms_all(self, indices, snaps=None, kind=1):
        """(R, len(indices), 3): the atoms `indices` of every member's snapshot (or live state): one gather, one read-back."""
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        out = np.zeros((len(self.engines), len(idx), 3))
        if self._lib.blues_batch_read_atoms(self._h, self._handles(snaps) if snaps is not None else None, int(kind),
                                            idx.ctypes.data_as(C.POINTER(C.c_int32)), len(idx), out.ctypes.data_as(_dp)):
            self._fail()
        return out

    def reset_all(self, active=None):
        for r, e in enumerate(self.engines):
            if active is None or active[r]:
                e.__dict__["_gcache"] = {}
        if self._lib.blues_batch_reset(self._h, self._mask(active)):
            self._fail()

    def set_velocities_to_temperature_all(self, temperature, seeds, active=None):
        R = len(self.engines)
        sd = (C.c_uint64 * R)(*[int(x) & 0xFFFFFFFFFFFFFFFF for x in seeds])
        if self._lib.blues_batch_set_velocities_to_temperature(self._h, float(temperature), sd, self._mask(active)):
            self._fail()

    def mesh_energy_all(self, with_alchemical_charges=False, active=None):
        """(R,) reciprocal-space mesh energies of the members' current coordinates: one launch, one read-back (blues_batch_mesh_energy)."""
        out = np.zeros(len(self.engines))
        if self._lib.blues_batch_mesh_energy(self._h, 1 if with_alchemical_charges else 0, self._mask(active), out.ctypes.data_as(_dp)):
            self._fail()
        return out

    def stats(self):
        s = (C.c_int64 * 4)()
        self._lib.blues_batch_get_stats(self._h, s)
        return {"lockstep_steps": s[0], "fallback_steps": s[1], "replicas": s[2], "batched_energy_evaluations": s[3]}

    COUNTERS = ("replans", "replan_seconds", "relayouts", "poll_resorts", "resort_seconds", "reshapes", "resorts", "tiles_per_list", "jcap", "nonbonded_kernel",
                "stragglers", "straggled", "rejoined", "straggle_seconds", "partial_steps")

    def counters(self):
        """What the batch's layout has cost so far (include/blues_engine.h: blues_batch_get_counters): re-plans of the layout shape and
        the seconds they took, members laid out again from a new sort, members re-sorted at the 64-step polls and the seconds that took,
        members moved to another shape in place; the shape now."""
        o = (C.c_double * _abi.N_BATCH_COUNTERS)()
        if self._lib.blues_batch_get_counters(self._h, o):
            raise EngineError(self._lib.blues_batch_last_error(self._h).decode())
        return {k: (float(o[i]) if k.endswith("seconds") else int(o[i])) for i, k in enumerate(self.COUNTERS)}

    def time_nonbonded(self, reps=20):
        u = C.c_double()
        if self._lib.blues_batch_time_nonbonded(self._h, int(reps), C.byref(u)):
            raise EngineError(self._lib.blues_batch_last_error(self._h).decode())
        return u.value

    def kernel_timing(self, every):
        """Bracket every `every`-th force launch of the nonbonded kernel in the stepping loop with HIP events (0: off); resets the statistics."""
        if self._lib.blues_batch_kernel_timing(self._h, int(every)):
            raise EngineError(self._lib.blues_batch_last_error(self._h).decode())

    def kernel_timing_result(self):
        """{"usec": mean duration of the sampled in-loop launches, "launches": how many, "usec_max": the longest}"""
        o = (C.c_double * 3)()
        if self._lib.blues_batch_get_kernel_timing(self._h, o):
            raise EngineError(self._lib.blues_batch_last_error(self._h).decode())
        return {"usec": o[0], "launches": int(o[1]), "usec_max": o[2]}

    def time_nonbonded_modes(self, reps=20):
        """(usec over current pruned lists, usec of a pass that re-derives them, share of the latter among the passes so far)"""
        u = (C.c_double * 2)(); f = C.c_double()
        if self._lib.blues_batch_time_nonbonded_modes(self._h, int(reps), u, C.byref(f)):
            raise EngineError(self._lib.blues_batch_last_error(self._h).decode())
        return u[0], u[1], f.value
