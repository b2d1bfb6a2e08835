import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blues_amd import build, integrators, systems
build.build_engine()
from blues_amd.engine import NativeEngine, NativeBatch
from blues_amd.replicas import replica_seed
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s, v = systems.s23k(mobile_atoms=275, frozen=True)
engs = []
for r in range(R):
    g = NativeEngine(s, integrators.generateNCMCIntegrator(nstepsNC=400, dt=0.004, temperature=300.0, seed=replica_seed(1234, r)).to_data(precision=0, replica=r))
    g.set_velocities(v); engs.append(g)
t0 = time.perf_counter(); B = NativeBatch(engs); print("batch create %.2f s" % (time.perf_counter() - t0), engs[0].stats())
for blk in range(int(os.environ.get("NBLK", "4"))):
    t0 = time.perf_counter(); B.step(50); dt = time.perf_counter() - t0
    print("block", blk, "%.1f us/round" % (1e6 * dt / 50), B.stats(), {k: engs[0].stats()[k] for k in ("jcap", "max_jcount", "resorts", "list_builds", "kernel_launches")}, flush=True)
