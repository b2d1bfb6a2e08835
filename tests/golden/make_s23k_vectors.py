#!/usr/bin/env python3
"""S23k golden vectors for the benchmarked configuration (BASELINE.json configs[0] to the letter, on the synthetic box):
run in the build container (a couple of minutes of CPU-oracle time), commit the output.

    python tests/golden/make_s23k_vectors.py

Inputs : blues_amd.systems.s23k(mobile_atoms=275, frozen=True) -- 23,400 atoms, 276 mobile, 15-atom alchemical toluene;
         generateNCMCIntegrator(nstepsNC=100, dt=4 fs, T=300 K, seed=SEED); velocities from the fixture.
Outputs: tests/golden/s23k_oracle_vectors.npz
  energies at 4 (lambda_sterics, lambda_electrostatics) pairs: total, the 8-term breakdown, forces on the mobile atoms;
  the 100-step protocol-work trace with a rigid rotation of the ligand about its centre of mass at step 50 (fixed rotation
  matrix + shift, stored), and the state (x, v of the mobile atoms -- frozen ones never move) every 10 steps, so a
  mixed-precision engine can be teacher-forced segment by segment where pointwise comparison over 0.4 ps is limited by
  the chaos of a liquid.

These are oracle (fp64 CPU restatement) outputs, NOT reference outputs: OpenMM cannot run here (oracle/blues_oracle.h).
They freeze the checker and give the GPU suite committed numbers for the configuration bench.py times.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from blues_amd import integrators, systems  # noqa: E402
from oracle import oracle  # noqa: E402

SEED = 20260
NSTEPS = 100
LAMBDAS = ((1.0, 1.0), (0.5, 0.3), (0.05, 0.0), (0.0, 0.0))
EVERY = 10


def rotation():
    """A fixed proper rotation (unit quaternion from a seeded draw) and a small shift."""
    q = np.random.RandomState(77).normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return R, np.array([0.013, -0.007, 0.004])


def apply_move(x, lig, mass, R, shift):
    """RandomLigandRotationMove.move (reference blues/moves.py:278-310): rotate about the mass-weighted centre."""
    com = (x[lig] * mass[lig, None]).sum(0) / mass[lig].sum()
    xn = x.copy()
    xn[lig] = (x[lig] - com) @ R.T + com + shift
    return xn


def main():
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    mob = np.nonzero(s.mass > 0)[0]
    lig = np.asarray(s.alchemical_atoms)
    data = integrators.generateNCMCIntegrator(nstepsNC=NSTEPS, dt=0.004, temperature=300.0, seed=SEED).to_data()
    o = oracle.Oracle(s, data); o.set_velocities(v)
    out = {"seed": SEED, "nsteps": NSTEPS, "dt": 0.004, "temperature": 300.0, "mobile_atoms": mob.astype(np.int32),
           "lambdas": np.array(LAMBDAS), "checkpoint_every": EVERY}
    t0 = time.time()
    E, T, F = [], [], []
    for ls, le in LAMBDAS:
        e, f, t = o.energy_forces(ls, le)
        E.append(e); T.append(t); F.append(f[mob])
        assert np.abs(f[mob]).max() > 0
    out["energy_total"] = np.array(E); out["energy_terms"] = np.array(T); out["forces_mobile"] = np.array(F)
    print("energies", E, "%.1f s" % (time.time() - t0))
    R, shift = rotation()
    out["move_rotation"] = R; out["move_shift"] = shift; out["move_step"] = NSTEPS // 2
    work, cx, cv, csteps = [], [], [], []

    def checkpoint(k):
        cx.append(o.get_positions()[mob]); cv.append(o.get_velocities()[mob]); csteps.append(k)
    checkpoint(0)
    for k in range(NSTEPS):
        if k == NSTEPS // 2:
            o.set_positions(apply_move(o.get_positions(), lig, s.mass, R, shift))
            out["x_after_move_mobile"] = o.get_positions()[mob]
        o.step(1)
        work.append(o.get_global("protocol_work"))
        if (k + 1) % EVERY == 0:
            checkpoint(k + 1)
            print("step", k + 1, "work", work[-1], "%.1f s" % (time.time() - t0))
    out["work_trace"] = np.array(work)
    out["checkpoint_steps"] = np.array(csteps, np.int32); out["checkpoint_x"] = np.array(cx); out["checkpoint_v"] = np.array(cv)
    path = os.path.join(ROOT, "tests", "golden", "s23k_oracle_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
