#!/usr/bin/env python3
"""S23k golden vectors for the two system VARIANTS bench.py runs beside the headline configuration (BASELINE.json configs[3] and
configs[4]; bench.py --workload water / sidechain): run in the build container (about ten minutes of CPU-oracle time), commit the output.

    python tests/golden/make_s23k_variant_vectors.py

solute    : blues_amd.systems.s23k_solute(frozen=True) -- the mobile region of the reference's freeze_radius: 285 SOLUTE atoms (the ligand + 18
            toluenes packed around it: bonds, angles, torsions, 1-4 exceptions, C-H constraint clusters), all water frozen; the move is a rigid
            rotation of the ligand at step 20 (python tests/golden/make_s23k_variant_vectors.py solute regenerates this variant only).
water     : blues_amd.systems.s23k(frozen=False, restrained=40) with the first water (atoms 15, 16, 17) alchemical -- every atom
            mobile, 40 position restraints (reference examples/example_water.py, blues/moves.py:846-1083: the species
            WaterTranslationMove inserts); the move of the vectors is a fixed translation of that water at step 20.
sidechain : blues_amd.systems.s23k(mobile_atoms=275, frozen=True) with atoms [0, 7, 8, 9] of the first toluene alchemical -- a
            PARTIALLY alchemical solute: alchemical-environment exclusions and 1-4 exceptions active (reference blues/moves.py:752-844,
            SideChainMove); the move is a fixed rotation of atoms 7, 8, 9 about the 1 -> 0 bond at step 20.

Per variant: total energy, the term breakdown and forces (water: on every 8th atom; sidechain: on the mobile atoms) at four
(lambda_sterics, lambda_electrostatics) pairs, the 40-step protocol-work trace, and the full state (x, v of the mobile atoms) at
step 20 AFTER the move and at step 40, so that a mixed-precision engine can be teacher-forced at the half-way point.

Oracle (fp64 CPU restatement) outputs, NOT reference outputs: OpenMM cannot run here (oracle/blues_oracle.h).
"""
import copy
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from blues_amd import integrators, systems  # noqa: E402
from oracle import oracle  # noqa: E402

NSTEPS = 40
MOVE_STEP = 20
LAMBDAS = ((1.0, 1.0), (0.5, 0.3), (0.05, 0.0), (0.0, 0.0))
SEEDS = {"water": 20261, "sidechain": 20262, "solute": 20263}


def variant(name):
    """(system, velocities, move(x) -> x') -- the same constructions as bench.py's build_chains."""
    if name == "water":
        base, vel = systems.s23k(frozen=False, restrained=40)
        s = copy.copy(base)
        s.alchemical_atoms = np.array([15, 16, 17], np.int32)
        shift = np.array([0.012, -0.017, 0.009])   # (a 40-step protocol cannot absorb a real translation: the water would land on its neighbours while it is switched back on)

        def move(x):
            xn = x.copy(); xn[[15, 16, 17]] += shift
            return xn
        return s, vel, move
    if name == "solute":
        # S23k-solute (tests/golden/make_s23k_solute.py): the ligand + 18 toluenes mobile -- 285 SOLUTE atoms with bonds, angles, torsions,
        # 1-4 exceptions and C-H constraint clusters -- ALL water frozen (reference blues/simulation.py:394-480: the mobile region of
        # freeze_radius); the move: a rigid rotation of the ligand by 0.35 rad about z through its centroid (RandomLigandRotationMove's geometry)
        s, vel = systems.s23k_solute(frozen=True)
        lig = np.arange(15)
        th = 0.35
        Rz = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])

        def move(x):
            xn = x.copy(); c = x[lig].mean(0); xn[lig] = c + (x[lig] - c) @ Rz.T
            return xn
        return s, vel, move
    base, vel = systems.s23k(mobile_atoms=275, frozen=True)
    s = copy.copy(base)
    s.alchemical_atoms = np.array([0, 7, 8, 9], np.int32)
    theta = 1.1

    def move(x):   # Rodrigues rotation of atoms 7, 8, 9 about the axis through atoms 1 -> 0 (reference blues/moves.py:731-750)
        a, b = x[1], x[0]
        k = (b - a) / np.linalg.norm(b - a)
        xn = x.copy()
        for i in (7, 8, 9):
            p = x[i] - b
            xn[i] = b + p * np.cos(theta) + np.cross(k, p) * np.sin(theta) + k * np.dot(k, p) * (1.0 - np.cos(theta))
        return xn
    return s, vel, move


def main():
    out = {"nsteps": NSTEPS, "move_step": MOVE_STEP, "dt": 0.004, "temperature": 300.0, "lambdas": np.array(LAMBDAS)}
    path = os.path.join(ROOT, "tests", "golden", "s23k_variant_vectors.npz")
    only = sys.argv[1:]       # (variants to (re)generate; the others are kept from the committed file)
    if only and os.path.exists(path):
        out.update({k: v for k, v in np.load(path).items() if not any(k.startswith(n + "_") for n in only)})
    for name in (only or ("water", "sidechain", "solute")):
        s, v, move = variant(name)
        mob = np.nonzero(s.mass > 0)[0]
        fsel = mob[::8] if name == "water" else mob
        data = integrators.generateNCMCIntegrator(nstepsNC=NSTEPS, dt=0.004, temperature=300.0, seed=SEEDS[name]).to_data()
        o = oracle.Oracle(s, data); o.set_velocities(v)
        t0 = time.time()
        E, T, F = [], [], []
        for ls, le in LAMBDAS:
            e, f, t = o.energy_forces(ls, le)
            E.append(e); T.append(t); F.append(f[fsel])
        print(name, "energies", E, "%.1f s" % (time.time() - t0), flush=True)
        work = []
        for k in range(NSTEPS):
            if k == MOVE_STEP:
                o.set_positions(move(o.get_positions()))
                xm, vm = o.get_positions()[mob].copy(), o.get_velocities()[mob].copy()
                wm = o.get_global("protocol_work")
            o.step(1)
            work.append(o.get_global("protocol_work"))
            if (k + 1) % 10 == 0:
                print(name, "step", k + 1, "work", work[-1], "%.1f s" % (time.time() - t0), flush=True)
        p = name + "_"
        out[p + "seed"] = SEEDS[name]; out[p + "alchemical_atoms"] = np.asarray(s.alchemical_atoms, np.int32)
        out[p + "mobile_atoms"] = mob.astype(np.int32); out[p + "force_atoms"] = fsel.astype(np.int32)
        out[p + "energy_total"] = np.array(E); out[p + "energy_terms"] = np.array(T); out[p + "forces"] = np.array(F)
        out[p + "work_trace"] = np.array(work); out[p + "work_before_move_step"] = wm
        out[p + "x_after_move"] = xm.astype(np.float64); out[p + "v_at_move"] = vm.astype(np.float64)
        out[p + "x_end"] = o.get_positions()[mob]
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
