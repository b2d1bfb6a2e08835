"""Do several replica batches on their own streams overlap usefully?  G batches of R/G chains stepped from G host threads."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blues_amd import build, integrators, systems
build.build_engine()
from blues_amd.engine import NativeEngine, NativeBatch
from blues_amd.replicas import replica_seed
system, vel = systems.s23k(mobile_atoms=275, frozen=True)
R = int(sys.argv[1]); nsteps = 400
engs = []
for r in range(R):
    integ = integrators.generateNCMCIntegrator(nstepsNC=nsteps + 100, dt=0.004, temperature=300.0, seed=replica_seed(1234, r))
    g = NativeEngine(system, integ.to_data(precision=0, replica=r)); g.set_velocities(vel); engs.append(g)
for G in [int(a) for a in sys.argv[2:]]:
    per = R // G
    batches = [NativeBatch(engs[k * per:(k + 1) * per]) for k in range(G)]
    for B in batches: B.step(20)
    def run(B): B.step(nsteps // 2 if False else 100)
    ths = [threading.Thread(target=run, args=(B,)) for B in batches]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print("R=%d in %d batches of %d: %.1f us per step of all chains, %.2f us/step/chain, %.0f ns/day" % (R, G, per, 1e6 * dt / 100, 1e6 * dt / 100 / R, R * 100 * 0.004e-3 / (dt / 86400.0)), flush=True)
    for B in batches: B.close()
