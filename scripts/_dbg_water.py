import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from blues_amd import build, integrators, systems
from blues_amd.engine import NativeEngine, NativeBatch
from blues_amd.replicas import replica_seed
system, vel = systems.s23k(frozen=False)
for R in (1, 2, 8):
    engs = []
    for r in range(R):
        integ = integrators.generateNCMCIntegrator(nstepsNC=150, dt=0.004, temperature=300.0, seed=replica_seed(1234, r))
        g = NativeEngine(system, integ.to_data(precision=0, replica=r)); g.set_velocities(vel); engs.append(g)
    B = NativeBatch(engs)
    for k in range(15):
        errs, _ = B.step(10, raise_errors=False)
        print(R, k, [str(e) for e in errs if e is not None][:2], [g.stats()["list_generation"] for g in engs][:4], "KE", engs[-1].kinetic_energy(), "W", engs[-1].get_global("protocol_work"), B.stats(), flush=True)
        if any(e is not None for e in errs): break
    B.close()
