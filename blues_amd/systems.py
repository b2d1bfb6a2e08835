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


def assemble_s23k_solute(base: SystemData, vel, removed_waters, inserted_x, inserted_v):
    """S23k with `len(inserted_x) / 15` more toluenes (the ligand's own parameters and topology) in place of the waters whose first atoms
    are `removed_waters`: atom order = ligand, inserted toluenes, everything else in the old order.  Returns (SystemData, velocities)."""
    n_lig = len(base.alchemical_atoms)
    lig = np.arange(n_lig)
    assert np.array_equal(np.asarray(base.alchemical_atoms), lig)
    n_new = len(inserted_x) // n_lig
    res = np.asarray(base.residue_of_atom)
    gone = np.zeros(base.n_atoms, bool)
    for a in removed_waters:
        gone[res == res[a]] = True
    keep = np.nonzero(~gone)[0]
    keep_rest = keep[keep >= n_lig]
    # old index -> new index (kept atoms); the ligand stays 0..n_lig-1, the inserted molecules follow
    new_of_old = np.full(base.n_atoms, -1, np.int64)
    new_of_old[lig] = lig
    new_of_old[keep_rest] = n_lig + n_new * n_lig + np.arange(len(keep_rest))
    n = n_lig + n_new * n_lig + len(keep_rest)

    def per_atom(a, new_vals):
        a = np.asarray(a)
        return np.concatenate([a[lig]] + [new_vals] * 1 + [a[keep_rest]])
    copies = lambda a: np.concatenate([np.asarray(a)[lig]] * n_new) if n_new else np.asarray(a)[:0]

    def terms(atoms, params):
        """rows of a term list re-indexed: rows of kept atoms, plus one copy per inserted molecule of the rows that lie inside the ligand"""
        atoms = np.asarray(atoms, np.int64).reshape(len(atoms), -1) if len(atoms) else np.zeros((0, 2), np.int64)
        params = np.asarray(params)
        if len(atoms) == 0:
            return atoms.astype(np.int32), params
        alive = ~gone[atoms].any(1)
        inside = (atoms < n_lig).all(1)
        out_a = [new_of_old[atoms[alive]]]
        out_p = [params[alive]] if params is not None and len(params) else []
        for m in range(n_new):
            out_a.append(atoms[inside] + n_lig * (m + 1))
            if params is not None and len(params):
                out_p.append(params[inside])
        return np.concatenate(out_a).astype(np.int32), (np.concatenate(out_p) if out_p else params)
    excl, _ = terms(base.exclusions, np.zeros((len(base.exclusions), 0)))
    exc_a, exc_p = terms(base.exception_atoms, base.exception_params)
    bond_a, bond_p = terms(base.bond_atoms, base.bond_params)
    ang_a, ang_p = terms(base.angle_atoms, base.angle_params)
    tor_a, tor_p = terms(base.torsion_atoms, base.torsion_params)
    con_a, con_d = terms(base.constraint_atoms, np.asarray(base.constraint_dist).reshape(-1, 1))
    pos = np.concatenate([base.positions[lig], np.asarray(inserted_x, np.float64).reshape(-1, 3), base.positions[keep_rest]])
    v = None
    if vel is not None:
        v = np.concatenate([vel[lig], np.asarray(inserted_v, np.float64).reshape(-1, 3), vel[keep_rest]])
    new_res = np.concatenate([np.zeros(n_lig, np.int64)] + [np.full(n_lig, m + 1, np.int64) for m in range(n_new)] + [np.unique(res[keep_rest], return_inverse=True)[1] + n_new + 1])
    s = SystemData(
        box=np.asarray(base.box, np.float64).copy(), mass=per_atom(base.mass, copies(base.mass)), charge=per_atom(base.charge, copies(base.charge)),
        sigma=per_atom(base.sigma, copies(base.sigma)), epsilon=per_atom(base.epsilon, copies(base.epsilon)),
        exclusions=excl, exception_atoms=exc_a, exception_params=exc_p, bond_atoms=bond_a, bond_params=bond_p, angle_atoms=ang_a, angle_params=ang_p,
        torsion_atoms=tor_a, torsion_params=tor_p, constraint_atoms=con_a, constraint_dist=np.asarray(con_d).reshape(-1),
        alchemical_atoms=np.arange(n_lig, dtype=np.int32), nonbonded_method=base.nonbonded_method, cutoff=base.cutoff, ewald_alpha=base.ewald_alpha,
        softcore_alpha=base.softcore_alpha, annihilate_electrostatics=base.annihilate_electrostatics, annihilate_sterics=base.annihilate_sterics,
        remove_cm_motion=base.remove_cm_motion, pme_order=base.pme_order, dispersion_correction=base.dispersion_correction, positions=pos,
        residue_of_atom=new_res.astype(np.int32))
    assert s.n_atoms == n
    return s, v


def s23k_solute(frozen=True):
    """S23k-solute: the S23k box with 18 more toluenes packed around the ligand in place of 90 waters (23,400 atoms still), built and
    relaxed by tests/golden/make_s23k_solute.py.  frozen=True: the ligand and those toluenes -- 285 atoms, all of them solute atoms with
    bonds, angles, torsions, 1-4 exceptions and C-H constraint clusters -- keep their mass, ALL water is frozen: the mobile region of
    the reference's `freeze_radius` (blues/simulation.py:394-480: `(<center> <: d) & !(<solvent>)`).  Returns (SystemData, velocities)."""
    z = np.load(os.path.join(_DATA, "s23k_solute.npz"))
    base, vel = s23k(frozen=False)
    s, v = assemble_s23k_solute(base, vel, z["removed_waters"], z["inserted_x"], z["inserted_v"])
    s.positions[z["moved_atoms"]] = z["moved_x"]
    if frozen:
        mobile = np.arange(len(base.alchemical_atoms) + len(z["inserted_x"]))
        s = freeze_except(s, mobile)
        v = v.copy(); v[s.mass == 0.0] = 0.0
    return s, v


# ---- the alchemical correction as a differential (reference blues/simulation.py:1100-1119)
ONE_4PI_EPS0 = 138.935456


def dispersion_correction_energy(system: SystemData, zero_epsilon=()):
    """OpenMM's long-range dispersion correction of a NonbondedForce with the cutoff and no switching function, in the form the engine
    and the oracle use (blues_engine.hip: pme_tables; oracle/blues_oracle.c: dispersion_correction) [recalled: NonbondedForceImpl::
    calcDispersionCorrection]; the atoms in `zero_epsilon` enter with epsilon 0 (disable_alchemical_dispersion_correction=True)."""
    n = system.n_atoms
    eps = np.array(system.epsilon, dtype=np.float64)
    eps[np.asarray(zero_epsilon, dtype=np.int64)] = 0.0
    sig = np.asarray(system.sigma, dtype=np.float64)
    classes, counts = np.unique(np.stack([sig, eps], axis=1), axis=0, return_counts=True)
    s12 = s6 = 0.0
    for a in range(len(classes)):
        for b in range(a, len(classes)):
            cnt = 0.5 * counts[a] * (counts[a] + 1.0) if a == b else float(counts[a]) * counts[b]
            sg = 0.5 * (classes[a][0] + classes[b][0]); e = np.sqrt(classes[a][1] * classes[b][1]); sg6 = sg ** 6
            s12 += cnt * e * sg6 * sg6; s6 += cnt * e * sg6
    tot = 0.5 * n * (n + 1.0)
    rc3 = system.cutoff ** 3; rc9 = rc3 ** 3
    V = float(np.prod(np.asarray(system.box, dtype=np.float64).reshape(-1)[:3]))
    return 8.0 * n * float(n) * np.pi * (s12 / tot / (9.0 * rc9) - s6 / tot / (3.0 * rc3)) / V


_FORCE_FIELD_ARRAYS = ("charge", "sigma", "epsilon", "exclusions", "exception_atoms", "exception_params", "bond_atoms", "bond_params", "angle_atoms",
                       "angle_params", "torsion_atoms", "torsion_params", "restraint_atoms", "restraint_x0")
_FORCE_FIELD_SCALARS = ("restraint_k", "nonbonded_method", "cutoff", "ewald_alpha", "softcore_alpha", "pme_order", "dispersion_correction")


def alchemical_difference_plan(alchemical: SystemData, plain: SystemData):
    """How D(x) = U_plain(x) - U_alchemical(x; lambda_sterics = lambda_electrostatics = 1) can be had WITHOUT evaluating U_plain, where
    `plain` is the non-alchemical System of the reference's md / alch contexts (blues/simulation.py:791-792) and `alchemical` the NCMC
    one.  If the two describe the same force field -- every parameter array equal; masses (freezing), constraints and the list of
    alchemical atoms are free to differ, they do not enter a potential energy -- the softcore forms at lambda = 1 ARE the plain ones
     and every direct-space term cancels.  What is left:
      * nothing, without reciprocal space ({"kind": "zero"});
      * with PME ('direct-space' alchemical treatment: the alchemical atoms' charges stay out of the mesh, the self term and the
        excluded-pair corrections, their epsilons out of the dispersion correction): the mesh energy with every atom's charge minus the
        mesh energy with the alchemical charges at 0 (the engine: blues_mesh_energy), the erf corrections of the excluded pairs that
        hold an alchemical atom (a handful of intra-ligand pairs: host arithmetic), and three constants of the box.
    Returns None when the Systems differ in anything else: the four energies of the reference's formula are evaluated then."""
    if alchemical is None or plain is None or alchemical.n_atoms != plain.n_atoms:
        return None
    if len(np.asarray(plain.alchemical_atoms).reshape(-1)) != 0 or len(np.asarray(alchemical.alchemical_atoms).reshape(-1)) == 0:
        return None
    if not np.array_equal(np.asarray(alchemical.box, dtype=np.float64).reshape(-1), np.asarray(plain.box, dtype=np.float64).reshape(-1)):
        return None
    for name in _FORCE_FIELD_ARRAYS:
        if not np.array_equal(np.asarray(getattr(alchemical, name)), np.asarray(getattr(plain, name))):
            return None
    for name in _FORCE_FIELD_SCALARS:
        if getattr(alchemical, name) != getattr(plain, name):
            return None
    if alchemical.nonbonded_method != NB_PME:
        return {"kind": "zero"}
    if tuple(alchemical.pme_grid) != tuple(plain.pme_grid) or getattr(alchemical, "barostat", None) or getattr(plain, "barostat", None):
        return None
    lig = np.asarray(alchemical.alchemical_atoms, dtype=np.int64)
    q = np.asarray(alchemical.charge, dtype=np.float64)
    is_lig = np.zeros(alchemical.n_atoms, bool); is_lig[lig] = True
    pairs = set()
    for arr in (alchemical.exclusions,):      # (the excluded pairs -- exceptions are among them -- as the engine and the oracle take them: blues_engine.hip T_EWEX)
        for a, b in np.asarray(arr, dtype=np.int64).reshape(-1, 2):
            if (is_lig[a] or is_lig[b]) and q[a] != 0.0 and q[b] != 0.0 and a != b:
                pairs.add((int(min(a, b)), int(max(a, b))))
    pairs = np.array(sorted(pairs), dtype=np.int64).reshape(-1, 2)
    atoms = np.unique(np.concatenate([lig, pairs.reshape(-1)]))          # whose coordinates the host part needs
    alpha = float(alchemical.ewald_alpha)
    V = float(np.prod(np.asarray(alchemical.box, dtype=np.float64).reshape(-1)[:3]))
    q_env = q.copy(); q_env[lig] = 0.0
    const = -ONE_4PI_EPS0 * alpha / np.sqrt(np.pi) * float((q[lig] ** 2).sum())
    const += -np.pi * ONE_4PI_EPS0 * (q.sum() ** 2 - q_env.sum() ** 2) / (2.0 * V * alpha * alpha)
    if alchemical.dispersion_correction:
        const += dispersion_correction_energy(alchemical) - dispersion_correction_energy(alchemical, zero_epsilon=lig)
    return {"kind": "pme", "atoms": atoms, "pairs": np.searchsorted(atoms, pairs), "qq": ONE_4PI_EPS0 * q[pairs[:, 0]] * q[pairs[:, 1]] if len(pairs) else np.zeros(0),
            "alpha": alpha, "box": np.asarray(alchemical.box, dtype=np.float64).reshape(-1)[:3].copy(), "const": float(const)}


def excluded_pair_term(plan, xyz):
    """- sum k_e q_i q_j erf(alpha r) / r over plan["pairs"], for coordinates xyz of plan["atoms"] ((..., n_atoms_of_plan, 3), nm)."""
    from scipy.special import erf
    xyz = np.asarray(xyz, dtype=np.float64)
    if len(plan["qq"]) == 0:
        return np.zeros(xyz.shape[:-2])
    d = xyz[..., plan["pairs"][:, 0], :] - xyz[..., plan["pairs"][:, 1], :]
    d -= plan["box"] * np.rint(d / plan["box"])
    r = np.sqrt((d * d).sum(-1))
    return -(plan["qq"] * erf(plan["alpha"] * r) / r).sum(-1)
