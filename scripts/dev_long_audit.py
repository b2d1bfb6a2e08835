"""List audit over a long all-mobile run (GPU box): a batch of S23k MD chains through the fragment lists, blues_audit_lists on two members
every few hundred steps -- across rebuilds, prunes, re-sorts by age and on request.  Prints the worst count of missing pairs (must be 0).
   python scripts/dev_long_audit.py [--R 8] [--steps 9000] [--every 300]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blues_amd import build, integrators, systems
build.build_engine()
from blues_amd.engine import NativeEngine, NativeBatch
from blues_amd.replicas import replica_seed, build_in_parallel

ap = argparse.ArgumentParser()
ap.add_argument("--R", type=int, default=8)
ap.add_argument("--steps", type=int, default=9000)
ap.add_argument("--every", type=int, default=300)
a = ap.parse_args()
s, vel = systems.s23k(frozen=False, restrained=0)
s.alchemical_atoms = np.zeros(0, np.int32)
engs = build_in_parallel(lambda r: NativeEngine(s, integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=replica_seed(5, r)).to_data(precision=0, replica=r)), a.R)
for g in engs:
    g.set_velocities(vel)
B = NativeBatch(engs)
worst, found = 0, 0
for done in range(0, a.steps, a.every):
    B.step(a.every)
    for g in (engs[0], engs[-1]):
        f, m = g.audit_lists(); worst = max(worst, m); found = f
st = engs[0].stats()
print("R=%d, %d steps, audited every %d: pairs inside the cutoff %d, worst missing %d; member 0: re-sorts %d rebuilds %d prunes %d; kinetic energy %.1f kJ/mol; fallback steps %d" % (
    a.R, a.steps, a.every, found, worst, st["resorts"], st["list_builds"], st["atom_prunes"], engs[0].kinetic_energy(), B.stats()["fallback_steps"]))
B.close()
for g in engs: g.close()
