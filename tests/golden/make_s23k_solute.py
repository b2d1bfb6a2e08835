#!/usr/bin/env python3
"""S23k-solute: an S23k variant whose mobile region looks like the reference's (missing #5).

`freeze_radius` keeps `(<center> <: d) & !(<solvent>)` (/root/reference/blues/simulation.py:394-480): the mobile atoms of a BLUES run
are SOLUTE atoms around the ligand -- bonds, angles, torsions, X-H constraint clusters -- and all water is frozen.  The headline S23k
configuration keeps the ligand and the 87 nearest rigid waters mobile instead (no bonded term, only water triangles in the constraint
solver).  This script builds the counterpart: 18 more toluenes (the ligand's own GAFF parameters, 15 atoms each: 270 atoms with
bonds / angles / torsions / 1-4 exceptions, a CH3 star and five C-H pairs per molecule in the constraint solver) packed around the
ligand in place of the 90 waters nearest to them -- the atom count stays 23,400 -- relaxed with the CPU oracle (steepest descent with
the constraints enforced, the environment beyond 1.7 nm of the ligand held fixed).

    python tests/golden/make_s23k_solute.py        (about ten minutes of CPU-oracle time; commit blues_amd/data/s23k_solute.npz)

What is stored: which waters were removed, and the relaxed coordinates and velocities of the inserted toluenes and of every atom that
moved; blues_amd.systems.s23k_solute() rebuilds the SystemData from the toluene-box fixture and this file.
"""
import copy
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from blues_amd import integrators, systems  # noqa: E402

N_TOL = 18
WATERS_PER_TOLUENE = 5
RELAX_RADIUS = 1.7


def rotation(rs):
    q = rs.normal(size=4); q /= np.linalg.norm(q)
    a, b, c, d = q
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                     [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])


def sites():
    """12 icosahedron vertices at 0.88 nm and 6 octahedron vertices at 1.30 nm around the ligand's centre."""
    phi = (1 + 5 ** 0.5) / 2
    ico = np.array([(0, s1, s2 * phi) for s1 in (1, -1) for s2 in (1, -1)] + [(s1, s2 * phi, 0) for s1 in (1, -1) for s2 in (1, -1)] + [(s2 * phi, 0, s1) for s1 in (1, -1) for s2 in (1, -1)], float)
    ico *= 0.88 / np.linalg.norm(ico[0])
    octa = 1.30 * np.array([(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)], float)
    # (a fixed rotation so that the two shells do not line up with each other or with the box axes)
    R = rotation(np.random.RandomState(11))
    return np.concatenate([ico, octa @ R.T])


def main():
    from oracle import oracle
    oracle.build()
    base, vel = systems.s23k(frozen=False)
    x = base.positions.copy()
    box = np.asarray(base.box, float).reshape(-1)[:3]
    res = np.asarray(base.residue_of_atom)
    lig = np.asarray(base.alchemical_atoms, np.int64)
    centre = x[lig].mean(0)
    shape = x[lig] - centre                                   # the ligand's own geometry: every inserted toluene is a rotated copy
    rs = np.random.RandomState(2026)
    new_x, new_v = [], []
    for c in sites():
        R = rotation(rs)
        new_x.append(centre + c + shape @ R.T)
        new_v.append(vel[lig] @ R.T)                          # (rotated with the molecule: the C-H constraints stay satisfied)
    new_x, new_v = np.concatenate(new_x), np.concatenate(new_v)
    # the waters that make room: for every inserted toluene the WATERS_PER_TOLUENE not yet removed whose oxygen is nearest to its atoms
    first = np.unique(res, return_index=True)[1]
    is_water = np.bincount(res)[res[first]] == 3
    w_first = first[is_water]
    removed = []
    taken = np.zeros(len(w_first), bool)
    for t in range(N_TOL):
        xt = new_x[15 * t:15 * t + 15]
        d = x[w_first][:, None, :] - xt[None, :, :]; d -= box * np.rint(d / box)
        dmin = np.sqrt((d * d).sum(-1)).min(1)
        dmin[taken] = 1e9
        pick = np.argsort(dmin)[:WATERS_PER_TOLUENE]
        taken[pick] = True
        removed.extend(int(w_first[p]) for p in pick)
    removed = np.array(sorted(removed))
    s = systems.assemble_s23k_solute(base, vel, removed, new_x, new_v)[0]
    print("atoms", s.n_atoms, "bonds", len(s.bond_atoms), "constraints", len(s.constraint_atoms), flush=True)
    # relax: everything within RELAX_RADIUS of the ligand's centre may move, the rest is held (mass 0: the oracle skips it)
    d = s.positions - centre; d -= box * np.rint(d / box)
    near = np.sqrt((d * d).sum(1)) < RELAX_RADIUS
    res2 = np.asarray(s.residue_of_atom)
    near = np.isin(res2, np.unique(res2[near]))               # whole molecules
    relax = copy.copy(s); relax.mass = np.where(near, s.mass, 0.0); relax.alchemical_atoms = np.zeros(0, np.int32)
    integ = integrators.generateNCMCIntegrator(nstepsNC=10, dt=0.001, temperature=300.0, seed=1)
    o = oracle.Oracle(relax, integ.to_data(precision=1), openmp=True)
    t0 = time.time()
    e0 = o.potential_energy()
    for rnd in range(12):
        e = o.minimize(50, 0.005)
        print("round %d: E = %.1f kJ/mol (start %.1f), %.0f s" % (rnd, e, e0, time.time() - t0), flush=True)
    xr = o.get_positions()
    moved = np.nonzero(np.abs(xr - s.positions).max(1) > 0)[0]
    print("atoms that moved:", len(moved), "largest displacement %.3f nm" % np.abs(xr - s.positions).max())
    out = os.path.join(ROOT, "blues_amd", "data", "s23k_solute.npz")
    np.savez_compressed(out, removed_waters=removed.astype(np.int32), inserted_x=xr[15:15 + 15 * N_TOL], inserted_v=new_v,
                        moved_atoms=moved.astype(np.int32), moved_x=xr[moved], energy=np.array([e0, e]))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
