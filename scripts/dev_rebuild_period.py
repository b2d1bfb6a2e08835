"""When do the members of a large batch rebuild their lists?  Bare stepping of a batch of the headline configuration (S23k, 276
mobile atoms, per-atom lists) with every member on its own displacement trigger (the default) and with every member rebuilding
whenever one asks (batch_sync_lists), rebuild counts beside the times.  Round 5 also tried, with a build that is not kept, members
rebuilding TOGETHER on every P-th step of the switch (an earlier threshold on those steps, `trig` on the others; a function of the
chain's own step index, so batch = solo still held): R = 1024, us per step: own triggers 694 (40.6 rebuilds per chain and 1000
steps); P = 4, 0.03 nm early: 739 (52.7); P = 4, 0.04: 748 (59.1); P = 4, 0.05: 775 (66.7); P = 8, 0.06: 768 (62.7); all together: 860
(102.5) -- the rebuild kernels are bound by their WORK (3.5 us per member that rebuilds), not by the latency of one member's chain:
bunching them buys nothing and the earlier thresholds cost rebuilds.
   python scripts/dev_rebuild_period.py [--R 1024] [--nsteps 400] ["batch_sync_lists=1" ...]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blues_amd import build, integrators, systems, tuning
build.build_engine()
from blues_amd.engine import NativeEngine, NativeBatch
from blues_amd.replicas import replica_seed, build_in_parallel

ap = argparse.ArgumentParser()
ap.add_argument("--R", type=int, default=1024)
ap.add_argument("--nsteps", type=int, default=400)
ap.add_argument("specs", nargs="*", default=["", "batch_sync_lists=1"])
a = ap.parse_args()
system, vel = systems.s23k(mobile_atoms=275, frozen=True)
for spec in a.specs:
    tuning.reset()
    tuning.set(assume_batch=a.R, **(tuning.parse(spec) if spec else {}))

    def make(r):
        integ = integrators.generateNCMCIntegrator(nstepsNC=a.nsteps + 300, dt=0.004, temperature=300.0, seed=replica_seed(1234, r))
        g = NativeEngine(system, integ.to_data(precision=0, replica=r))
        g.set_velocities(vel * (1.0 + 0.001 * (r % 7)))
        return g
    engs = build_in_parallel(make, a.R)
    B = NativeBatch(engs)
    B.step(200)
    s0 = [g.stats()["list_builds"] for g in engs[:64]]
    t0 = time.perf_counter(); B.step(a.nsteps); dt = time.perf_counter() - t0
    s1 = [g.stats()["list_builds"] for g in engs[:64]]
    k1 = B.time_nonbonded(20)
    st = B.stats()
    au = engs[-1].audit_lists()
    print("[%-40s] R=%d: %.1f us per step (%.0f ns/day); rebuilds per chain and 1000 steps %.1f; K1 alone %.1f us; lockstep %d fallback %d; mode %d; audit %s" % (
        spec, a.R, 1e6 * dt / a.nsteps, a.R * a.nsteps * 0.004e-3 / (dt / 86400.0), 1000.0 * (np.sum(s1) - np.sum(s0)) / 64.0 / a.nsteps, k1,
        st["lockstep_steps"], st["fallback_steps"], engs[0].stats()["nonbonded_kernel"], au), flush=True)
    B.close()
    for g in engs: g.close()
tuning.reset()
