"""ctypes mirror of include/blues_engine.h (the C-ABI structs and prototypes).

Both the product library (libblues_hip.so) and the test-only CPU oracle
(oracle/_build/libblues_oracle.so) consume the same two descriptor structs, so
this module only describes the ABI; it never loads the oracle.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

ABI_VERSION = 6
SWITCH_NONE, SWITCH_VV, SWITCH_GHMC = 0, 1, 2
NB_NOCUTOFF = 0
NB_PME_DIRECT = 1
NB_PME = 2
N_ENERGY_TERMS = 10
N_STATS = 22
N_BATCH_COUNTERS = 15
ENERGY_TERM_NAMES = ("bonds", "angles", "torsions", "nonbonded", "exceptions", "alch_sterics",
                     "alch_electrostatics", "restraint", "reciprocal", "dispersion_correction")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class BluesSystemDesc(C.Structure):
    _fields_ = [
        ("n_atoms", C.c_int32),
        ("box", C.c_double * 9),
        ("mass", _dp), ("charge", _dp), ("sigma", _dp), ("epsilon", _dp),
        ("n_exclusions", C.c_int32), ("exclusions", _ip),
        ("n_exceptions", C.c_int32), ("exception_atoms", _ip), ("exception_params", _dp),
        ("n_bonds", C.c_int32), ("bond_atoms", _ip), ("bond_params", _dp),
        ("n_angles", C.c_int32), ("angle_atoms", _ip), ("angle_params", _dp),
        ("n_torsions", C.c_int32), ("torsion_atoms", _ip), ("torsion_params", _dp),
        ("n_constraints", C.c_int32), ("constraint_atoms", _ip), ("constraint_dist", _dp),
        ("n_alchemical", C.c_int32), ("alchemical_atoms", _ip),
        ("n_restraints", C.c_int32), ("restraint_atoms", _ip), ("restraint_x0", _dp), ("restraint_k", C.c_double),
        ("nonbonded_method", C.c_int32),
        ("cutoff", C.c_double), ("ewald_alpha", C.c_double), ("softcore_alpha", C.c_double),
        ("annihilate_electrostatics", C.c_int32), ("annihilate_sterics", C.c_int32),
        ("remove_cm_motion", C.c_int32),
        ("pme_grid", C.c_int32 * 3), ("pme_order", C.c_int32), ("dispersion_correction", C.c_int32),
    ]


class BluesIntegratorDesc(C.Structure):
    _fields_ = [
        ("timestep", C.c_double), ("temperature", C.c_double), ("collision_rate", C.c_double),
        ("nsteps_neq", C.c_int32), ("nprop", C.c_int32),
        ("prop_lambda_min", C.c_double), ("prop_lambda_max", C.c_double),
        ("splitting", C.c_char_p),
        ("n_lambda_steps", C.c_int32),
        ("lambda_sterics", _dp), ("lambda_electrostatics", _dp),
        ("constraint_tolerance", C.c_double),
        ("seed", C.c_uint64), ("replica", C.c_int32), ("precision", C.c_int32),
        ("switching_mode", C.c_int32), ("steps_per_propagation", C.c_int32),
    ]


class BluesTuning(C.Structure):
    """include/blues_engine.h: BluesTuning (launch-policy overrides; the library reads no environment variables)."""
    _fields_ = [
        ("struct_size", C.c_int32), ("plain_skin", C.c_int32),
        ("skin", C.c_double), ("prune_margin", C.c_double), ("jcap_scale", C.c_double), ("acap_scale", C.c_double),
        ("k1_mode", C.c_int32), ("list_group", C.c_int32), ("sub_iw", C.c_int32), ("sub_chunks", C.c_int32),
        ("seg_len", C.c_int32), ("waves_per_block", C.c_int32), ("k2_jiter", C.c_int32),
        ("fuse_forces", C.c_int32), ("fuse_big", C.c_int32), ("fast_step", C.c_int32), ("slot_mask", C.c_int32),
        ("fork", C.c_int32), ("use_graph", C.c_int32), ("graph_units", C.c_int32), ("graph_fork", C.c_int32),
        ("batch_sync_lists", C.c_int32), ("force_lists", C.c_int32), ("no_sphere", C.c_int32),
        ("pme_general", C.c_int32), ("debug_lists", C.c_int32), ("assume_batch", C.c_int32), ("k1_threads", C.c_int32),
        ("k2_dense", C.c_int32), ("k2_early", C.c_int32), ("fuse_finalize", C.c_int32), ("host_threads", C.c_int32),
        ("pack_clusters", C.c_int32),
    ]


def _f64(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _i32(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int32))
    if shape is not None:
        a = a.reshape(shape)
    return a


@dataclass
class SystemData:
    """Host-side description of the alchemical System (numpy, OpenMM units).

    Field meanings follow BluesSystemDesc in include/blues_engine.h.
    """
    box: np.ndarray                      # (3,) orthorhombic edge lengths, nm
    mass: np.ndarray
    charge: np.ndarray
    sigma: np.ndarray
    epsilon: np.ndarray
    exclusions: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.int32))
    exception_atoms: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.int32))
    exception_params: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    bond_atoms: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.int32))
    bond_params: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))
    angle_atoms: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.int32))
    angle_params: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))
    torsion_atoms: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.int32))
    torsion_params: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    constraint_atoms: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.int32))
    constraint_dist: np.ndarray = field(default_factory=lambda: np.zeros((0,)))
    alchemical_atoms: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    restraint_atoms: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    restraint_x0: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    restraint_k: float = 0.0
    nonbonded_method: int = NB_PME_DIRECT
    cutoff: float = 1.0
    ewald_alpha: float = 0.0
    softcore_alpha: float = 0.5
    annihilate_electrostatics: bool = True
    annihilate_sterics: bool = False
    remove_cm_motion: bool = False
    pme_grid: tuple = (0, 0, 0)          # NB_PME: mesh of the reciprocal sum
    pme_order: int = 5
    dispersion_correction: bool = True   # NB_PME: OpenMM's NonbondedForce default
    barostat: tuple = None               # (pressure [bar], temperature [K], frequency): MonteCarloBarostat of the MD leg (host-side, blues_amd/barostat.py)
    positions: np.ndarray = None         # (n,3) nm, optional initial coordinates
    residue_of_atom: np.ndarray = None   # optional bookkeeping for host-side selections
    names: list = None
    extras: dict = None                  # oracle-only test extras (custom forces of the ethylene known-answer system); ignored by the engine

    @property
    def n_atoms(self):
        return int(len(self.mass))

    def to_desc(self):
        """Returns (BluesSystemDesc, keepalive) -- keepalive owns the numpy buffers."""
        n = self.n_atoms
        keep = {}
        d = BluesSystemDesc()
        d.n_atoms = n
        box = np.asarray(self.box, dtype=np.float64).reshape(-1)
        if box.size == 3:
            box9 = np.zeros(9); box9[0], box9[4], box9[8] = box
        else:
            box9 = box
        for i in range(9):
            d.box[i] = float(box9[i])

        def put_f(name, arr, shape=None):
            a = _f64(arr, shape); keep[name] = a
            setattr(d, name, a.ctypes.data_as(_dp))
            return a

        def put_i(name, arr, shape=None):
            a = _i32(arr, shape); keep[name] = a
            setattr(d, name, a.ctypes.data_as(_ip))
            return a

        put_f("mass", self.mass, (n,)); put_f("charge", self.charge, (n,))
        put_f("sigma", self.sigma, (n,)); put_f("epsilon", self.epsilon, (n,))
        d.n_exclusions = len(put_i("exclusions", self.exclusions, (-1, 2)))
        d.n_exceptions = len(put_i("exception_atoms", self.exception_atoms, (-1, 2)))
        put_f("exception_params", self.exception_params, (-1, 3))
        d.n_bonds = len(put_i("bond_atoms", self.bond_atoms, (-1, 2))); put_f("bond_params", self.bond_params, (-1, 2))
        d.n_angles = len(put_i("angle_atoms", self.angle_atoms, (-1, 3))); put_f("angle_params", self.angle_params, (-1, 2))
        d.n_torsions = len(put_i("torsion_atoms", self.torsion_atoms, (-1, 4))); put_f("torsion_params", self.torsion_params, (-1, 3))
        d.n_constraints = len(put_i("constraint_atoms", self.constraint_atoms, (-1, 2))); put_f("constraint_dist", self.constraint_dist, (-1,))
        d.n_alchemical = len(put_i("alchemical_atoms", self.alchemical_atoms, (-1,)))
        d.n_restraints = len(put_i("restraint_atoms", self.restraint_atoms, (-1,))); put_f("restraint_x0", self.restraint_x0, (-1, 3))
        d.restraint_k = float(self.restraint_k)
        d.nonbonded_method = int(self.nonbonded_method)
        d.cutoff = float(self.cutoff); d.ewald_alpha = float(self.ewald_alpha); d.softcore_alpha = float(self.softcore_alpha)
        d.annihilate_electrostatics = int(bool(self.annihilate_electrostatics))
        d.annihilate_sterics = int(bool(self.annihilate_sterics))
        d.remove_cm_motion = int(bool(self.remove_cm_motion))
        for k in range(3):
            d.pme_grid[k] = int(self.pme_grid[k])
        d.pme_order = int(self.pme_order); d.dispersion_correction = int(bool(self.dispersion_correction))
        return d, keep


@dataclass
class IntegratorData:
    """Numeric form of AlchemicalExternalLangevinIntegrator's constructor arguments."""
    timestep: float                      # ps
    temperature: float                   # K
    nsteps_neq: int
    lambda_sterics: np.ndarray           # [n_lambda_steps+1]
    lambda_electrostatics: np.ndarray
    splitting: str = "H V R O R V H"
    collision_rate: float = 1.0
    nprop: int = 1
    prop_lambda_min: float = 0.2
    prop_lambda_max: float = 0.8
    constraint_tolerance: float = 1e-8
    seed: int = 0
    replica: int = 0
    precision: int = 0                   # 0 mixed, 1 double
    switching_mode: int = 0              # SWITCH_NONE / SWITCH_VV / SWITCH_GHMC (reference blues/switching.py)
    steps_per_propagation: int = 1

    @property
    def n_lambda_steps(self):
        if self.switching_mode:
            return int(self.nsteps_neq)
        return int(self.nsteps_neq) * self.splitting.count("H")

    def to_desc(self):
        keep = {}
        d = BluesIntegratorDesc()
        d.timestep = float(self.timestep); d.temperature = float(self.temperature)
        d.collision_rate = float(self.collision_rate)
        d.nsteps_neq = int(self.nsteps_neq); d.nprop = int(self.nprop)
        d.prop_lambda_min = float(self.prop_lambda_min); d.prop_lambda_max = float(self.prop_lambda_max)
        keep["splitting"] = self.splitting.encode()
        d.splitting = keep["splitting"]
        nl = self.n_lambda_steps
        d.n_lambda_steps = nl
        ls = _f64(self.lambda_sterics, (-1,)); le = _f64(self.lambda_electrostatics, (-1,))
        if len(ls) != nl + 1 or len(le) != nl + 1:
            raise ValueError("lambda tables must have n_lambda_steps+1 = %d entries" % (nl + 1))
        keep["ls"], keep["le"] = ls, le
        d.lambda_sterics = ls.ctypes.data_as(_dp); d.lambda_electrostatics = le.ctypes.data_as(_dp)
        d.constraint_tolerance = float(self.constraint_tolerance)
        d.seed = int(self.seed) & 0xFFFFFFFFFFFFFFFF; d.replica = int(self.replica); d.precision = int(self.precision)
        d.switching_mode = int(self.switching_mode); d.steps_per_propagation = int(self.steps_per_propagation)
        return d, keep


def declare_engine_prototypes(lib):
    """Attach argtypes/restype for every symbol include/blues_engine.h declares."""
    H = C.c_void_p
    protos = {
        "blues_engine_create": ([C.POINTER(BluesSystemDesc), C.POINTER(BluesIntegratorDesc), C.c_int, C.POINTER(H)], C.c_int),
        "blues_engine_destroy": ([H], C.c_int),
        "blues_last_error": ([H], C.c_char_p),
        "blues_abi_version": ([], C.c_int),
        "blues_tuning_default": ([C.POINTER(BluesTuning)], None),
        "blues_set_tuning": ([C.POINTER(BluesTuning)], C.c_int),
        "blues_get_tuning": ([C.POINTER(BluesTuning)], C.c_int),
        "blues_set_positions": ([H, _dp, C.c_int32], C.c_int),
        "blues_set_velocities": ([H, _dp, C.c_int32], C.c_int),
        "blues_set_box": ([H, _dp], C.c_int),
        "blues_get_positions": ([H, _dp, C.c_int32], C.c_int),
        "blues_get_velocities": ([H, _dp, C.c_int32], C.c_int),
        "blues_get_forces": ([H, _dp, C.c_int32], C.c_int),
        "blues_get_box": ([H, _dp], C.c_int),
        "blues_set_velocities_to_temperature": ([H, C.c_double, C.c_uint64], C.c_int),
        "blues_get_energy": ([H, _dp, _dp], C.c_int),
        "blues_get_energy_at": ([H, C.c_double, C.c_double, _dp], C.c_int),
        "blues_get_energy_terms": ([H, _dp], C.c_int),
        "blues_mesh_energy": ([H, C.c_int32, _dp], C.c_int),
        "blues_step": ([H, C.c_int32], C.c_int),
        "blues_run_switch": ([H, C.c_int32, _dp], C.c_int),
        "blues_get_global": ([H, C.c_char_p, _dp], C.c_int),
        "blues_set_global": ([H, C.c_char_p, C.c_double], C.c_int),
        "blues_reset": ([H], C.c_int),
        "blues_get_stats": ([H, C.POINTER(C.c_int64)], C.c_int),
        "blues_time_nonbonded": ([H, C.c_int32, _dp], C.c_int),
        "blues_time_list_build": ([H, C.c_int32, _dp], C.c_int),
        "blues_audit_lists": ([H, C.POINTER(C.c_int64)], C.c_int),
        "blues_snapshot_capture": ([H, C.c_int32, C.POINTER(H)], C.c_int),
        "blues_snapshot_release": ([H], C.c_int),
        "blues_snapshot_read": ([H, C.c_int32, _dp, C.c_int32], C.c_int),
        "blues_set_positions_from_snapshot": ([H, H], C.c_int),
        "blues_set_velocities_from_snapshot": ([H, H], C.c_int),
        "blues_snapshot_read_atoms": ([H, C.c_int32, C.POINTER(C.c_int32), C.c_int32, _dp], C.c_int),
        "blues_set_positions_from_snapshot_edited": ([H, H, C.POINTER(C.c_int32), C.c_int32, _dp], C.c_int),
        "blues_batch_create": ([C.POINTER(H), C.c_int32, C.POINTER(H)], C.c_int),
        "blues_batch_destroy": ([H], C.c_int),
        "blues_batch_last_error": ([H], C.c_char_p),
        "blues_batch_size": ([H], C.c_int),
        "blues_batch_step": ([H, C.c_int32, _dp, C.POINTER(C.c_int32)], C.c_int),
        "blues_batch_set_active": ([H, C.POINTER(C.c_int32)], C.c_int),
        "blues_batch_prefetch_energies": ([H, C.c_int32], C.c_int),
        "blues_batch_snapshot_capture": ([H, C.c_int32, C.POINTER(C.c_int32), C.POINTER(H)], C.c_int),
        "blues_batch_restore": ([H, C.POINTER(H), C.c_int32], C.c_int),
        "blues_batch_restore_edited": ([H, C.POINTER(H), C.POINTER(C.c_int32), C.c_int32, _dp], C.c_int),
        "blues_batch_read_atoms": ([H, C.POINTER(H), C.c_int32, C.POINTER(C.c_int32), C.c_int32, _dp], C.c_int),
        "blues_batch_reset": ([H, C.POINTER(C.c_int32)], C.c_int),
        "blues_batch_set_velocities_to_temperature": ([H, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)], C.c_int),
        "blues_batch_mesh_energy": ([H, C.c_int32, C.POINTER(C.c_int32), _dp], C.c_int),
        "blues_batch_get_stats": ([H, C.POINTER(C.c_int64)], C.c_int),
        "blues_batch_get_counters": ([H, _dp], C.c_int),
        "blues_batch_time_nonbonded": ([H, C.c_int32, _dp], C.c_int),
        "blues_batch_time_nonbonded_modes": ([H, C.c_int32, _dp, _dp], C.c_int),
        "blues_debug_setup_seconds": ([_dp], C.c_int),
        "blues_debug_check_guards": ([C.POINTER(C.c_int64)], C.c_int),
        "blues_batch_kernel_timing": ([H, C.c_int32], C.c_int),
        "blues_batch_get_kernel_timing": ([H, _dp], C.c_int),
    }
    for name, (args, res) in protos.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.argtypes = args
        fn.restype = res
    return sorted(protos)


ENGINE_SYMBOLS = (
    "blues_engine_create", "blues_engine_destroy", "blues_last_error", "blues_abi_version",
    "blues_tuning_default", "blues_set_tuning", "blues_get_tuning",
    "blues_set_positions", "blues_set_velocities", "blues_set_box", "blues_get_positions",
    "blues_get_velocities", "blues_get_forces", "blues_get_box", "blues_set_velocities_to_temperature",
    "blues_get_energy", "blues_get_energy_at", "blues_get_energy_terms", "blues_mesh_energy", "blues_step", "blues_run_switch", "blues_get_global",
    "blues_set_global", "blues_reset", "blues_get_stats", "blues_time_nonbonded", "blues_time_list_build", "blues_audit_lists",
    "blues_snapshot_capture", "blues_snapshot_release", "blues_snapshot_read", "blues_set_positions_from_snapshot",
    "blues_set_velocities_from_snapshot", "blues_snapshot_read_atoms", "blues_set_positions_from_snapshot_edited",
    "blues_batch_create", "blues_batch_destroy", "blues_batch_last_error", "blues_batch_size", "blues_batch_step", "blues_batch_set_active", "blues_batch_prefetch_energies",
    "blues_batch_snapshot_capture", "blues_batch_restore", "blues_batch_restore_edited", "blues_batch_read_atoms", "blues_batch_reset",
    "blues_batch_set_velocities_to_temperature", "blues_batch_mesh_energy",
    "blues_batch_get_stats", "blues_batch_get_counters", "blues_batch_time_nonbonded", "blues_batch_time_nonbonded_modes",
    "blues_batch_kernel_timing", "blues_batch_get_kernel_timing", "blues_debug_setup_seconds", "blues_debug_check_guards",
)
