"""How the fragment-list path ages with its sort order (GPU box): a batch of all-mobile S23k chains stepped for several thousand steps;
per segment the step time, the nonbonded kernel alone, re-sorts so far.
   python scripts/dev_order_age.py [--R 16] [--segments 8] [--seg-steps 1000]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blues_amd import build, integrators, systems
build.build_engine()
from blues_amd.engine import NativeEngine, NativeBatch
from blues_amd.replicas import replica_seed, build_in_parallel

ap = argparse.ArgumentParser()
ap.add_argument("--R", type=int, default=16)
ap.add_argument("--segments", type=int, default=8)
ap.add_argument("--seg-steps", type=int, default=1000)
a = ap.parse_args()
s, vel = systems.s23k(frozen=False, restrained=0)
s.alchemical_atoms = np.zeros(0, np.int32)


def make(r):
    g = NativeEngine(s, integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=replica_seed(77, r)).to_data(precision=0, replica=r))
    g.set_velocities(vel)
    return g


engs = build_in_parallel(make, a.R)
B = NativeBatch(engs)
B.step(100)
for seg in range(a.segments):
    t0 = time.perf_counter(); B.step(a.seg_steps); dt = time.perf_counter() - t0
    k1 = B.time_nonbonded(10)
    st = engs[0].stats()
    print("steps %5d: %.1f us per step (%.2f per chain-step), K1 alone %.1f us (%.2f per chain), member 0: re-sorts %d rebuilds %d prunes %d" % (
        100 + (seg + 1) * a.seg_steps, 1e6 * dt / a.seg_steps, 1e6 * dt / a.seg_steps / a.R, k1, k1 / a.R, st.get("resorts", -1), st["list_builds"], st["atom_prunes"]), flush=True)
B.close()
for g in engs: g.close()
