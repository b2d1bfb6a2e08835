"""Forces of the per-atom-list kernel (a lone engine laid out as a batch member) against the lone-engine layout, at the initial
geometry and after a few steps: python scripts/dev_forcecheck.py [spec ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blues_amd import build, integrators, systems, tuning
from blues_amd.engine import NativeEngine
system, vel = systems.s23k(mobile_atoms=275, frozen=True)
mob = np.nonzero(system.mass > 0)[0]
def eng():
    g = NativeEngine(system, integrators.generateNCMCIntegrator(nstepsNC=100, dt=0.004, temperature=300.0, seed=5).to_data(precision=0)); g.set_velocities(vel); return g
tuning.reset()
ref = eng(); f0 = ref.get_forces()[mob]
for spec in (sys.argv[1:] or ["assume_batch=8"]):
    tuning.reset(); tuning.set(**tuning.parse(spec))
    g = eng()
    f = g.get_forces()[mob]
    d = np.abs(f - f0); i = np.unravel_index(np.argmax(d), d.shape)
    print("[%s] kernel %d pruned %d: first evaluation max|df| %.3e (|f|max %.1f) at mobile atom %d (%s); nan %d" % (spec, g.stats()["nonbonded_kernel"], g.stats()["pruned_lists"], d.max(), np.abs(f0).max(), i[0], f[i[0]] - f0[i[0]], int(np.isnan(f).sum())), flush=True)
    f2 = g.get_forces()[mob]     # second evaluation: pruned lists in use
    d2 = np.abs(f2 - f0)
    print("      second evaluation max|df| %.3e; atoms off by > 1e-2: %s" % (d2.max(), np.nonzero(d2.max(1) > 1e-2)[0][:20]), flush=True)
    g.close()
