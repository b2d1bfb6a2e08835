"""Nonbonded-kernel study on a GPU box: for each tuning spec, a replica batch of R S23k chains is stepped and the kernel is
timed in both kinds of pass (over current pruned lists / re-deriving them), with the list statistics that explain the times.
   python scripts/dev_k1.py [--R 256] [--nsteps 200] "skin=0.16,prune_margin=0.04" "prune_margin=0" ..."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blues_amd import build, integrators, systems, tuning
build.build_engine()
from blues_amd.engine import NativeEngine, NativeBatch
from blues_amd.replicas import replica_seed

ap = argparse.ArgumentParser()
ap.add_argument("--R", type=int, default=256)
ap.add_argument("--nsteps", type=int, default=200)
ap.add_argument("--warm", type=int, default=60)
ap.add_argument("specs", nargs="*", default=[""])
a = ap.parse_args()
system, vel = systems.s23k(mobile_atoms=275, frozen=True)
for spec in a.specs:
    tuning.reset()
    if spec:
        tuning.set(**tuning.parse(spec))
    engs = []
    for r in range(a.R):
        integ = integrators.generateNCMCIntegrator(nstepsNC=a.nsteps + a.warm + 10, dt=0.004, temperature=300.0, seed=replica_seed(1234, r))
        g = NativeEngine(system, integ.to_data(precision=0, replica=r)); g.set_velocities(vel); engs.append(g)
    B = NativeBatch(engs)
    B.step(a.warm)
    s0 = [g.stats() for g in engs[:8]]
    t0 = time.perf_counter(); B.step(a.nsteps); dt = time.perf_counter() - t0
    s1 = [g.stats() for g in engs[:8]]
    u0, u1, f = B.time_nonbonded_modes(30)
    nat = 261.0
    reb = np.mean([b["list_generation"] - a_["list_generation"] for a_, b in zip(s0, s1)]) / a.nsteps
    pr = np.mean([b["atom_prunes"] - a_["atom_prunes"] for a_, b in zip(s0, s1)]) / a.nsteps / nat
    st = s1[0]
    algo = 36.0 * system.n_atoms * a.R
    blend = (1 - pr) * u0 + pr * u1
    print("[%s] R=%d step %.1f us  (%.0f ns/day) | K1 pruned %.1f us, prune pass %.1f us, prune share %.3f (history %.3f) -> mean %.1f us = %.1f%% of 8 TB/s | rebuilds/step %.3f | lists: outer %.0f/atom (%.2f chunks) pruned %.0f/atom (%.2f chunks) jcap %d max_jcount %d S=%d mode %d"
          % (spec, a.R, 1e6 * dt / a.nsteps, a.R * a.nsteps * 0.004e-3 / (dt / 86400.0), u0, u1, pr, f, blend, 100 * algo / blend / 1e3 / 8000.0, reb,
             st["atom_list_entries"] / nat, st["atom_list_iterations"] / nat, st["pruned_list_entries"] / nat, st["pruned_list_iterations"] / nat, st["jcap"], st["max_jcount"], st["tiles_per_list"], st["nonbonded_kernel"]), flush=True)
    B.close()
    for g in engs: g.close()
