"""Times the batched nonbonded kernel at the initial geometry (no stepping) -- for experiment builds of the library
   (BLUES_LIB_PATH=...) and tuning specs: python scripts/dev_k1exp.py [R] [spec ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blues_amd import build, integrators, systems, tuning
from blues_amd.engine import NativeEngine, NativeBatch
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
system, vel = systems.s23k(mobile_atoms=275, frozen=True)
for spec in (sys.argv[2:] or [""]):
    tuning.reset()
    if spec:
        tuning.set(**tuning.parse(spec))
    engs = []
    for r in range(R):
        g = NativeEngine(system, integrators.generateNCMCIntegrator(nstepsNC=100, dt=0.004, temperature=300.0, seed=r).to_data(precision=0, replica=r)); g.set_velocities(vel); engs.append(g)
    B = NativeBatch(engs)
    u0, u1, f = B.time_nonbonded_modes(30)
    st = engs[0].stats()
    print("%s [%s] R=%d: pruned-list pass %.1f us, full-list pass %.1f us | pruned %.2f chunks/atom, full %.2f" % (os.path.basename(build.LIB_PATH), spec, R, u0, u1, st["pruned_list_iterations"] / 261.0, st["atom_list_iterations"] / 261.0), flush=True)
    B.close()
    for g in engs: g.close()
