"""Runs the list audit (blues_audit_lists) at every step of a hot S23k chain; BLUES_LIB_PATH selects the library (a build with the
first version's mobile prune margin shows what the audit catches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BLUES_TUNING", "assume_batch=512")
from blues_amd import systems, integrators
from blues_amd.engine import NativeEngine
s, v = systems.s23k(mobile_atoms=275, frozen=True)
g = NativeEngine(s, integrators.generateNCMCIntegrator(nstepsNC=200, dt=0.004, temperature=450.0, seed=11).to_data(precision=0)); g.set_velocities(1.5 * v)
tot = miss = 0; bad_steps = 0
for step in range(150):
    g.step(1)
    f, m = g.audit_lists()
    tot += f; miss += m; bad_steps += m > 0
st = g.stats()
print("pairs in range (summed over steps)", tot, "missing", miss, "steps with a miss", bad_steps, "| builds", st["list_builds"], "prunes", st["atom_prunes"])
