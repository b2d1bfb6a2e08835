"""Aggregate NCMC throughput of R replicas sharing launches on one GPU (S23k flagship workload).
   python scripts/batch_scaling.py [--workload rotmove|water] [--nsteps 400] R1 R2 ..."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blues_amd import build, integrators, systems
build.build_engine()
from blues_amd.engine import NativeEngine, NativeBatch
from blues_amd.replicas import replica_seed

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="rotmove")
ap.add_argument("--nsteps", type=int, default=400)
ap.add_argument("--pme", action="store_true")
ap.add_argument("R", nargs="*", type=int, default=[1, 2, 4, 8, 16, 32])
a = ap.parse_args()
system, vel = systems.s23k(frozen=False) if a.workload == "water" else systems.s23k(mobile_atoms=275, frozen=True)
if a.pme:
    system = systems.with_reciprocal_space(system)
rng = np.random.RandomState(3)
for R in a.R:
    engs = []
    for r in range(R):
        integ = integrators.generateNCMCIntegrator(nstepsNC=a.nsteps + 50, dt=0.004, temperature=300.0, seed=replica_seed(1234, r))
        g = NativeEngine(system, integ.to_data(precision=0, replica=r))
        g.set_velocities(vel)
        engs.append(g)
    B = NativeBatch(engs)
    B.step(50)
    t0 = time.perf_counter(); B.step(a.nsteps); dt = time.perf_counter() - t0
    k1 = B.time_nonbonded(20)
    st = B.stats()
    ns_day = R * a.nsteps * 0.004e-3 / (dt / 86400.0)
    algo = 36.0 * system.n_atoms * R
    print("R=%3d  %.1f us/step-round  %.2f us/step/replica  aggregate %.0f ns/day | K1 batched %.1f us -> %.0f GB/s algorithmic (%.2f%% of 8 TB/s) | lockstep %d fallback %d rebuilds %d"
          % (R, 1e6 * dt / a.nsteps, 1e6 * dt / a.nsteps / R, ns_day, k1, algo / k1 / 1e3, 100 * algo / k1 / 1e3 / 8000.0, st["lockstep_steps"], st["fallback_steps"], engs[0].stats()["list_generation"]), flush=True)
    B.close()
    for g in engs: g.close()
