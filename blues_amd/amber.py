"""Minimal Amber prmtop / inpcrd readers and the Amber -> OpenMM-unit conversions.

BLUES loads its inputs with ParmEd (reference blues/settings.py:60-90,
`parmed.load_file(prmtop, xyz=inpcrd)`) and turns them into an OpenMM System
with `structure.createSystem(**kwargs)` (reference blues/simulation.py:219).
Neither ParmEd nor OpenMM exists on the MI355X box, so this module restates the
small part of that conversion the switching path needs: per-atom charge / LJ /
mass, bonded terms, exclusions and 1-4 exceptions, HBonds + rigid-water
constraints and hydrogen-mass repartitioning, all in OpenMM units
(nm, kJ/mol, amu, e).  Conversions: SURVEY.md Appendix B.
"""
import re

import numpy as np

from ._abi import NB_PME, NB_PME_DIRECT, SystemData

AMBER_CHARGE = 18.2223
KCAL = 4.184
WATER_NAMES = ("WAT", "HOH", "TIP3", "TP3", "SPC")


def _parse_format(fmt):
    m = re.match(r"\(?(\d+)([aAiIeEfF])(\d+)(?:\.(\d+))?\)?", fmt.strip())
    if not m:
        raise ValueError("unsupported prmtop format %r" % fmt)
    return int(m.group(1)), m.group(2).lower(), int(m.group(3))


def read_prmtop(path):
    """Returns {FLAG: list or ndarray}."""
    flags = {}
    name, fmt, buf = None, None, []

    def flush():
        if name is None:
            return
        _, kind, width = fmt
        items = []
        for line in buf:
            line = line.rstrip("\n")
            for k in range(0, len(line), width):
                tok = line[k:k + width]
                if tok.strip() == "" and kind != "a":
                    continue
                items.append(tok)
        if kind == "i":
            flags[name] = np.array([int(t) for t in items], dtype=np.int64)
        elif kind in "ef":
            flags[name] = np.array([float(t) for t in items], dtype=np.float64)
        else:
            flags[name] = [t.strip() for t in items if t != ""]

    with open(path) as fh:
        for line in fh:
            if line.startswith("%VERSION") or line.startswith("%COMMENT"):
                continue
            if line.startswith("%FLAG"):
                flush()
                name, fmt, buf = line.split()[1], None, []
            elif line.startswith("%FORMAT"):
                fmt = _parse_format(line[len("%FORMAT"):])
            else:
                buf.append(line)
        flush()
    return flags


def read_inpcrd(path):
    """Returns (positions nm (n,3), velocities or None, box nm (3,) or None)."""
    with open(path) as fh:
        lines = fh.read().split("\n")
    natom = int(lines[1].split()[0])
    vals = []
    for line in lines[2:]:
        for k in range(0, len(line.rstrip()), 12):
            tok = line[k:k + 12].strip()
            if tok:
                vals.append(float(tok))
    vals = np.array(vals)
    pos = vals[:3 * natom].reshape(natom, 3) * 0.1
    rest = vals[3 * natom:]
    vel, box = None, None
    if len(rest) >= 3 * natom:
        vel = rest[:3 * natom].reshape(natom, 3) * 0.1 * 20.455  # Amber time unit -> nm/ps
        rest = rest[3 * natom:]
    if len(rest) >= 3:
        box = rest[:3] * 0.1
    return pos, vel, box


def read_rst7(path):
    """Amber restart (.rst7 / .rst) in either form parmed.amber.Rst7 accepts (reference blues/settings.py:76-85): the NetCDF
    "AMBERRESTART" file the reference's own RestartReporter writes (blues/reporters.py:217-225, netcdf=True), recognised by its
    magic bytes, or the ASCII inpcrd layout with a velocity block.  Returns (positions nm, velocities nm/ps or None, box nm or None)."""
    from .formats import AmberNetCDFRestart
    if AmberNetCDFRestart.is_netcdf(path):
        pos, vel, box, _ = AmberNetCDFRestart.read(path)
        return pos, vel, box
    return read_inpcrd(path)


def write_rst7(path, positions_nm, velocities_nm_ps=None, box_nm=None, title="written by blues_amd", time_ps=0.0, netcdf=False):
    """Amber restart: NetCDF (netcdf=True, the reference's choice) or ASCII in the fixed 6F12.7 layout (angstrom; velocities in
    angstrom per 1/20.455 ps)."""
    if netcdf:
        from .formats import AmberNetCDFRestart
        return AmberNetCDFRestart.write(path, positions_nm, velocities_nm_ps, box_nm, time_ps=time_ps, title=title)
    x = np.asarray(positions_nm, dtype=np.float64).reshape(-1, 3) * 10.0
    blocks = [x.reshape(-1)]
    if velocities_nm_ps is not None:
        blocks.append(np.asarray(velocities_nm_ps, dtype=np.float64).reshape(-1) * 10.0 / 20.455)
    with open(path, "w") as fh:
        fh.write(title[:80] + "\n")
        fh.write("%5d%15.7e\n" % (len(x), time_ps) if velocities_nm_ps is not None else "%5d\n" % len(x))
        for b in blocks:
            for k in range(0, len(b), 6):
                fh.write("".join("%12.7f" % v for v in b[k:k + 6]) + "\n")
        if box_nm is not None:
            fh.write("".join("%12.7f" % v for v in list(np.asarray(box_nm, dtype=np.float64).reshape(-1)[:3] * 10.0) + [90.0, 90.0, 90.0]) + "\n")


def ewald_alpha(cutoff, tolerance):
    """OpenMM: alpha = sqrt(-ln(2 tol)) / cutoff (2.145966 for 0.005 at 1 nm)."""
    return float(np.sqrt(-np.log(2.0 * tolerance)) / cutoff)


def system_from_amber(prm, positions, box, cutoff=1.0, ewald_error_tolerance=0.005, constraints="HBonds",
                      rigid_water=True, hydrogen_mass=None, remove_cm_motion=True, alchemical_atoms=(),
                      tip3p_for_untyped_water=True, reciprocal_space=True, dispersion_correction=True):
    """Amber topology -> SystemData, following structure.createSystem's kwargs
    (reference examples/rotmove_cuda.yml:19-27: PME, 10 A cutoff, HBonds,
    rigidWater, removeCMMotion, hydrogenMass 3.024, ewaldErrorTolerance 0.005).
    Constrained bonds and rigid-water angles are left out of the harmonic terms
    (they contribute zero at the constrained geometry).
    reciprocal_space=True is nonbondedMethod=PME in full (mesh of OpenMM's Reference platform for the tolerance, self term,
    excluded-pair corrections, dispersion correction); False keeps the direct-space sum only and says so in the log --
    forces on the water then differ from the reference's, only the protocol work (lambda-dependent pairs) does not."""
    p = prm["POINTERS"]
    natom, ntypes = int(p[0]), int(p[1])
    charge = prm["CHARGE"] / AMBER_CHARGE
    mass = prm["MASS"].copy()
    atnum = prm.get("ATOMIC_NUMBER")
    if atnum is None:
        atnum = np.where(mass < 1.5, 1, 6)
    tidx = prm["ATOM_TYPE_INDEX"]
    nbidx = prm["NONBONDED_PARM_INDEX"]
    acoef, bcoef = prm["LENNARD_JONES_ACOEF"], prm["LENNARD_JONES_BCOEF"]
    res_ptr = list(prm["RESIDUE_POINTER"] - 1) + [natom]
    res_lab = prm["RESIDUE_LABEL"]
    residue_of_atom = np.zeros(natom, dtype=np.int32)
    for r in range(len(res_lab)):
        residue_of_atom[res_ptr[r]:res_ptr[r + 1]] = r
    is_water_res = np.array([lab in WATER_NAMES for lab in res_lab])
    is_water = is_water_res[residue_of_atom]

    sigma = np.zeros(natom)
    eps = np.zeros(natom)
    for i in range(natom):
        t = int(tidx[i])
        if t <= 0:
            if not (tip3p_for_untyped_water and is_water[i]):
                raise ValueError("atom %d has no LJ type" % i)
            if atnum[i] == 8:  # TIP3P oxygen
                sigma[i], eps[i] = 0.315075, 0.635968
            else:
                sigma[i], eps[i] = 0.1, 0.0
            continue
        k = int(nbidx[ntypes * (t - 1) + (t - 1)]) - 1
        a, b = acoef[k], bcoef[k]
        if a == 0.0 or b == 0.0:
            sigma[i], eps[i] = 0.1, 0.0
        else:
            sigma[i] = (a / b) ** (1.0 / 6.0) * 0.1
            eps[i] = b * b / (4.0 * a) * KCAL

    def triples(name, width):
        arr = prm.get(name)
        if arr is None or len(arr) == 0:
            return np.zeros((0, width), dtype=np.int64)
        return np.asarray(arr).reshape(-1, width)

    bonds = np.vstack([triples("BONDS_INC_HYDROGEN", 3), triples("BONDS_WITHOUT_HYDROGEN", 3)])
    angles = np.vstack([triples("ANGLES_INC_HYDROGEN", 4), triples("ANGLES_WITHOUT_HYDROGEN", 4)])
    dihs = np.vstack([triples("DIHEDRALS_INC_HYDROGEN", 5), triples("DIHEDRALS_WITHOUT_HYDROGEN", 5)])
    bk, br = prm["BOND_FORCE_CONSTANT"], prm["BOND_EQUIL_VALUE"]
    ak, at = prm["ANGLE_FORCE_CONSTANT"], prm["ANGLE_EQUIL_VALUE"]
    dk, dn, dp = prm["DIHEDRAL_FORCE_CONSTANT"], prm["DIHEDRAL_PERIODICITY"], prm["DIHEDRAL_PHASE"]
    scee = prm.get("SCEE_SCALE_FACTOR", np.full(len(dk), 1.2))
    scnb = prm.get("SCNB_SCALE_FACTOR", np.full(len(dk), 2.0))

    # hydrogen mass repartitioning (createSystem(hydrogenMass=...))
    if hydrogen_mass is not None:
        for b in bonds:
            i, j = int(b[0]) // 3, int(b[1]) // 3
            if atnum[i] == 1:
                i, j = j, i
            if atnum[j] == 1 and atnum[i] != 1:
                transfer = hydrogen_mass - mass[j]
                mass[j] = hydrogen_mass
                mass[i] -= transfer

    cons_atoms, cons_dist = [], []
    bond_atoms, bond_params = [], []
    constrained_pair = set()
    for b in bonds:
        i, j, t = int(b[0]) // 3, int(b[1]) // 3, int(b[2]) - 1
        r0, k = br[t] * 0.1, 2.0 * bk[t] * KCAL * 100.0
        has_h = atnum[i] == 1 or atnum[j] == 1
        water = is_water[i] and is_water[j]
        if (constraints in ("HBonds", "AllBonds", "HAngles") and has_h) or constraints in ("AllBonds", "HAngles") or (rigid_water and water):
            cons_atoms.append((i, j)); cons_dist.append(r0)
            constrained_pair.add((min(i, j), max(i, j)))
        else:
            bond_atoms.append((i, j)); bond_params.append((r0, k))
    angle_atoms, angle_params = [], []
    for a in angles:
        i, j, k_, t = int(a[0]) // 3, int(a[1]) // 3, int(a[2]) // 3, int(a[3]) - 1
        th0, kk = at[t], 2.0 * ak[t] * KCAL
        if rigid_water and is_water[i] and is_water[j] and is_water[k_]:
            pair = (min(i, k_), max(i, k_))
            if pair not in constrained_pair:  # H-H distance from the two O-H bonds and the angle
                d_ij = next(d for (p_, d) in zip(cons_atoms, cons_dist) if set(p_) == {i, j})
                d_kj = next(d for (p_, d) in zip(cons_atoms, cons_dist) if set(p_) == {k_, j})
                cons_atoms.append((i, k_)); cons_dist.append(float(np.sqrt(d_ij ** 2 + d_kj ** 2 - 2 * d_ij * d_kj * np.cos(th0))))
                constrained_pair.add(pair)
            continue
        angle_atoms.append((i, j, k_)); angle_params.append((th0, kk))
    tors_atoms, tors_params = [], []
    exc = {}
    for d in dihs:
        i, j, k_, l, t = int(d[0]) // 3, int(d[1]) // 3, abs(int(d[2])) // 3, abs(int(d[3])) // 3, int(d[4]) - 1
        if dk[t] != 0.0:
            tors_atoms.append((i, j, k_, l)); tors_params.append((round(abs(dn[t])), dp[t], dk[t] * KCAL))
        if int(d[2]) >= 0 and int(d[3]) >= 0:  # carries a 1-4 interaction
            pair = (min(i, l), max(i, l))
            if pair not in exc:
                qq = charge[i] * charge[l] / (scee[t] if scee[t] != 0 else 1.2)
                e14 = np.sqrt(eps[i] * eps[l]) / (scnb[t] if scnb[t] != 0 else 2.0)
                exc[pair] = (qq, 0.5 * (sigma[i] + sigma[l]), e14)

    excl = set()
    nex, exl = prm["NUMBER_EXCLUDED_ATOMS"], prm["EXCLUDED_ATOMS_LIST"]
    pos_ = 0
    for i in range(natom):
        for q in range(int(nex[i])):
            j = int(exl[pos_ + q]) - 1
            if j >= 0:
                excl.add((min(i, j), max(i, j)))
        pos_ += int(nex[i])
    excl |= set(exc.keys()) | constrained_pair
    excl = np.array(sorted(excl), dtype=np.int32).reshape(-1, 2)
    exc_pairs = sorted(exc.keys())

    alpha = ewald_alpha(cutoff, ewald_error_tolerance)
    if reciprocal_space:
        from .systems import pme_grid_for
        method, grid = NB_PME, pme_grid_for(box, alpha, cutoff, ewald_error_tolerance)
    else:
        import logging
        logging.getLogger(__name__).warning("system_from_amber: PME lowered to its direct-space part (reciprocal_space=False): no mesh, self or "
                                            "dispersion terms; energies and environment forces are not those of nonbondedMethod=PME")
        method, grid = NB_PME_DIRECT, (0, 0, 0)
    return SystemData(
        pme_grid=grid, pme_order=5, dispersion_correction=bool(dispersion_correction),
        box=np.asarray(box, dtype=np.float64), mass=mass, charge=charge, sigma=sigma, epsilon=eps,
        exclusions=excl,
        exception_atoms=np.array(exc_pairs, dtype=np.int32).reshape(-1, 2),
        exception_params=np.array([exc[p_] for p_ in exc_pairs], dtype=np.float64).reshape(-1, 3),
        bond_atoms=np.array(bond_atoms, dtype=np.int32).reshape(-1, 2), bond_params=np.array(bond_params).reshape(-1, 2),
        angle_atoms=np.array(angle_atoms, dtype=np.int32).reshape(-1, 3), angle_params=np.array(angle_params).reshape(-1, 2),
        torsion_atoms=np.array(tors_atoms, dtype=np.int32).reshape(-1, 4), torsion_params=np.array(tors_params, dtype=np.float64).reshape(-1, 3),
        constraint_atoms=np.array(cons_atoms, dtype=np.int32).reshape(-1, 2), constraint_dist=np.array(cons_dist, dtype=np.float64),
        alchemical_atoms=np.array(sorted(alchemical_atoms), dtype=np.int32),
        nonbonded_method=method, cutoff=cutoff, ewald_alpha=alpha,
        remove_cm_motion=remove_cm_motion, positions=np.asarray(positions, dtype=np.float64),
        residue_of_atom=residue_of_atom, names=list(prm.get("ATOM_NAME", [])),
    )
