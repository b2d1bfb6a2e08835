"""Host-side System manipulation: freezing, restraints, tiling, and the synthetic
~23k-atom benchmark box (S23k).

Mirrors the *semantics* of SystemFactory.freeze_atoms / freeze_radius /
restrain_positions (reference blues/simulation.py:319-480): frozen atoms get
mass 0 (reference blues/utils.py:202-221) and are skipped by the integrator but
still exert forces; restraints add k*periodicdistance(x,x0)^2.
The Amber-mask machinery itself is out of scope; selections are index arrays.
"""
import copy
import os

import numpy as np

from ._abi import NB_PME, NB_PME_DIRECT, SystemData

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_ARRAY_FIELDS = ("box", "mass", "charge", "sigma", "epsilon", "exclusions", "exception_atoms", "exception_params",
                 "bond_atoms", "bond_params", "angle_atoms", "angle_params", "torsion_atoms", "torsion_params",
                 "constraint_atoms", "constraint_dist", "alchemical_atoms", "restraint_atoms", "restraint_x0",
                 "positions", "residue_of_atom")
_SCALAR_FIELDS = ("restraint_k", "nonbonded_method", "cutoff", "ewald_alpha", "softcore_alpha",
                  "annihilate_electrostatics", "annihilate_sterics", "remove_cm_motion")
_OPTIONAL_SCALARS = ("pme_order", "dispersion_correction")   # (files written before reciprocal space existed do not have them)


def save_system(path, system: SystemData, **extra):
    d = {k: np.asarray(getattr(system, k)) for k in _ARRAY_FIELDS if getattr(system, k) is not None}
    d.update({k: np.asarray(getattr(system, k)) for k in _SCALAR_FIELDS + _OPTIONAL_SCALARS})
    d["pme_grid"] = np.asarray(system.pme_grid, dtype=np.int32)
    d.update({k: np.asarray(v) for k, v in extra.items()})
    np.savez_compressed(path, **d)


def load_system(path):
    z = np.load(path)
    kw = {k: z[k] for k in _ARRAY_FIELDS if k in z.files}
    for k in _SCALAR_FIELDS:
        v = z[k].item()
        kw[k] = v
    for k in _OPTIONAL_SCALARS:
        if k in z.files:
            kw[k] = z[k].item()
    if "pme_grid" in z.files:
        kw["pme_grid"] = tuple(int(v) for v in z["pme_grid"])
    extra = {k: z[k] for k in z.files if k not in kw}
    return SystemData(**kw), extra


def ewald_tolerance(alpha, cutoff):
    """Inverse of alpha = sqrt(-ln(2 tol)) / cutoff (OpenMM NonbondedForceImpl::calcPMEParameters)."""
    return 0.5 * np.exp(-(alpha * cutoff) ** 2)


def pme_grid_for(box, alpha, cutoff, tol=None):
    """Mesh OpenMM's Reference platform uses for ewaldErrorTolerance tol: ceil(2 alpha L / (3 tol^(1/5))) per edge, at least 6
    [recalled: OpenMM 7.4.2 NonbondedForceImpl::calcPMEParameters; the GPU platforms round up further to FFT-friendly sizes]."""
    if tol is None:
        tol = ewald_tolerance(alpha, cutoff)
    return tuple(max(6, int(np.ceil(2.0 * alpha * float(L) / (3.0 * tol ** 0.2) - 1e-9))) for L in np.asarray(box, dtype=np.float64).reshape(-1)[:3])


def with_reciprocal_space(system: SystemData, dispersion_correction=True):
    """The same system under nonbondedMethod=PME in full (reference blues/simulation.py:219 with examples/rotmove_cuda.yml:20):
    direct space as before plus the reciprocal-space mesh, self term, excluded-pair corrections and the dispersion correction."""
    s = copy.copy(system)
    s.nonbonded_method = NB_PME
    s.pme_grid = pme_grid_for(system.box, system.ewald_alpha, system.cutoff)
    s.pme_order = 5
    s.dispersion_correction = bool(dispersion_correction)
    return s


def add_barostat(system: SystemData, temperature, pressure_bar=1.0, frequency=25):
    """SimulationFactory.addBarostat (reference blues/simulation.py:603-626): MonteCarloBarostat(pressure, temperature, frequency) on the MD system."""
    s = copy.copy(system)
    s.barostat = (float(pressure_bar), float(temperature), int(frequency))
    return s


def freeze_atoms(system: SystemData, frozen_idx):
    """freeze_atoms / utils.zero_masses (reference blues/simulation.py:364-392)."""
    s = copy.copy(system)
    s.mass = np.array(system.mass, dtype=np.float64, copy=True)
    s.mass[np.asarray(frozen_idx, dtype=np.int64)] = 0.0
    return s


def freeze_except(system: SystemData, mobile_idx):
    mask = np.ones(system.n_atoms, dtype=bool)
    mask[np.asarray(mobile_idx, dtype=np.int64)] = False
    return freeze_atoms(system, np.nonzero(mask)[0])


def min_image(d, box):
    return d - box * np.round(d / box)


def freeze_radius(system: SystemData, center_idx, distance_nm, solvent_mask=None):
    """freeze_radius (reference blues/simulation.py:394-480): every atom NOT in
    (whole residues with an atom within `distance_nm` of the centre selection) minus
    the solvent selection is frozen."""
    x = system.positions
    res = system.residue_of_atom
    if res is None:
        res = np.arange(system.n_atoms)
    near = np.zeros(system.n_atoms, dtype=bool)
    for c in np.asarray(center_idx):
        d = min_image(x - x[c], system.box)
        near |= (d * d).sum(1) <= distance_nm ** 2
    mobile_res = np.unique(res[near])
    mobile = np.isin(res, mobile_res)
    if solvent_mask is not None:
        mobile &= ~np.asarray(solvent_mask, dtype=bool)
    mobile[np.asarray(center_idx)] = True
    return freeze_except(system, np.nonzero(mobile)[0])


def restrain_positions(system: SystemData, atom_idx, weight_kj_per_nm2):
    """restrain_positions (reference blues/simulation.py:319-362): weight in kJ/mol/nm^2
    (5 kcal/mol/A^2 from YAML = 2092 kJ/mol/nm^2)."""
    s = copy.copy(system)
    s.restraint_atoms = np.asarray(atom_idx, dtype=np.int32)
    s.restraint_x0 = np.array(system.positions[np.asarray(atom_idx)], dtype=np.float64)
    s.restraint_k = float(weight_kj_per_nm2)
    return s


def tile_system(system: SystemData, reps):
    """Replicates a periodic box reps=(nx,ny,nz) times (independent copies of every term)."""
    nx, ny, nz = reps
    n = system.n_atoms
    ncopy = nx * ny * nz

    def rep_idx(a):
        a = np.asarray(a)
        if a.size == 0:
            return a
        return np.concatenate([a + c * n for c in range(ncopy)], axis=0).astype(np.int32)

    def rep_val(a):
        a = np.asarray(a)
        if a.size == 0:
            return a
        return np.concatenate([a] * ncopy, axis=0)

    shifts = np.array([(i, j, k) for i in range(nx) for j in range(ny) for k in range(nz)], dtype=np.float64) * system.box
    pos = np.concatenate([system.positions + s for s in shifts], axis=0)
    res = system.residue_of_atom
    nres = int(res.max()) + 1 if res is not None else 0
    return SystemData(
        box=system.box * np.array(reps, dtype=np.float64),
        mass=rep_val(system.mass), charge=rep_val(system.charge), sigma=rep_val(system.sigma), epsilon=rep_val(system.epsilon),
        exclusions=rep_idx(system.exclusions), exception_atoms=rep_idx(system.exception_atoms), exception_params=rep_val(system.exception_params),
        bond_atoms=rep_idx(system.bond_atoms), bond_params=rep_val(system.bond_params),
        angle_atoms=rep_idx(system.angle_atoms), angle_params=rep_val(system.angle_params),
        torsion_atoms=rep_idx(system.torsion_atoms), torsion_params=rep_val(system.torsion_params),
        constraint_atoms=rep_idx(system.constraint_atoms), constraint_dist=rep_val(system.constraint_dist),
        alchemical_atoms=np.asarray(system.alchemical_atoms, dtype=np.int32),  # first copy only
        restraint_atoms=np.zeros((0,), np.int32), restraint_x0=np.zeros((0, 3)), restraint_k=0.0,
        nonbonded_method=system.nonbonded_method, cutoff=system.cutoff, ewald_alpha=system.ewald_alpha,
        softcore_alpha=system.softcore_alpha, annihilate_electrostatics=system.annihilate_electrostatics,
        annihilate_sterics=system.annihilate_sterics, remove_cm_motion=system.remove_cm_motion,
        pme_order=system.pme_order, dispersion_correction=system.dispersion_correction,
        pme_grid=pme_grid_for(system.box * np.array(reps, dtype=np.float64), system.ewald_alpha, system.cutoff) if system.nonbonded_method == NB_PME else (0, 0, 0),
        positions=pos,
        residue_of_atom=None if res is None else np.concatenate([res + c * nres for c in range(ncopy)]).astype(np.int32),
        names=None if system.names is None else list(system.names) * ncopy,
    )


def toluene_box():
    """The 975-atom toluene-in-water box (15-atom GAFF toluene + 320 TIP3P waters, cubic
    2.1786 nm) derived from the reference's blues/tests/data/TOL-parm.{prmtop,inpcrd} fixture
    by tests/golden/make_fixtures.py, relaxed and thermalised at 300 K.
    Returns (SystemData, velocities)."""
    s, extra = load_system(os.path.join(_DATA, "tol_box.npz"))
    return s, extra.get("velocities")


def nearest_molecules(system: SystemData, center_idx, n_atoms_target, exclude_idx=()):
    """Whole residues ordered by distance of their first atom to the centroid of `center_idx`."""
    x = system.positions
    c = x[np.asarray(center_idx)].mean(0)
    res = system.residue_of_atom
    first = np.unique(res, return_index=True)[1]
    d = min_image(x[first] - c, system.box)
    order = np.argsort((d * d).sum(1), kind="stable")
    chosen = []
    excl = set(int(i) for i in exclude_idx)
    for r in order:
        atoms = np.nonzero(res == res[first[r]])[0]
        if excl & set(int(a) for a in atoms):
            continue
        chosen.extend(int(a) for a in atoms)
        if len(chosen) >= n_atoms_target:
            break
    return np.array(chosen, dtype=np.int64)


def s23k(mobile_atoms=275, reps=(2, 3, 4), frozen=True, restrained=0, restraint_weight=2092.0):
    """S23k: the toluene box tiled 2x3x4 -> 23,400 atoms, 4.3572 x 6.5358 x 8.7144 nm.
    The first toluene is the alchemical ligand.  With frozen=True only the ligand and the
    nearest whole molecules (about `mobile_atoms` atoms in all) keep their mass, emulating
    freeze_radius 5 A (reference examples/rotmove_cuda.yml:42-45; 275 mobile atoms in
    docs/BLUES_tutorial.ipynb:718).  Returns (SystemData, velocities)."""
    base, vel = toluene_box()
    big = tile_system(base, reps)
    ncopy = reps[0] * reps[1] * reps[2]
    v = None if vel is None else np.concatenate([vel] * ncopy, axis=0)
    lig = np.asarray(big.alchemical_atoms, dtype=np.int64)
    if restrained:
        # a fixed subset of heavy atoms stands in for the backbone restraint of examples/water_cuda.yaml:36-38
        heavy = np.nonzero(big.mass > 10.0)[0]
        big = restrain_positions(big, heavy[:: max(1, len(heavy) // restrained)][:restrained], restraint_weight)
    if frozen:
        extra = nearest_molecules(big, lig, mobile_atoms - len(lig), exclude_idx=lig)
        mobile = np.concatenate([lig, extra])
        big = freeze_except(big, mobile)
        if v is not None:
            v = v.copy(); v[big.mass == 0.0] = 0.0
    return big, v
