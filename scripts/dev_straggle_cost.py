"""What a member's leaving and coming back costs the batch at the benchmark's scale (2048 engines in the process, a batch of 1024):
   python scripts/dev_straggle_cost.py [R]"""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from blues_amd import build, integrators, systems, tuning
build.build_engine()
from blues_amd.engine import NativeEngine, NativeBatch
from blues_amd.replicas import replica_seed, build_in_parallel
from test_gpu_batch import _swap_waters_outward
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
s, v = systems.s23k(mobile_atoms=275, frozen=True)
tuning.set(assume_batch=R)
def make(r):
    g = NativeEngine(s, integrators.generateNCMCIntegrator(nstepsNC=1000, dt=0.004, temperature=300.0, seed=replica_seed(1234, r)).to_data(precision=0, replica=r)); g.set_velocities(v * (1.0 + 1e-4 * r)); return g
engs = build_in_parallel(make, 2 * R)
B = NativeBatch(engs[:R])
last = {}
def timed(label, n=20):
    t0 = time.perf_counter(); B.step(n); dt = time.perf_counter() - t0
    c = B.counters(); st = B.stats()
    print("%-44s %7.1f ms for %d steps | straggled %d rejoined %d stragglers %d straggle_seconds %.4f fallback %d partial %d resorts %d" % (label, 1e3 * dt, n, c["straggled"], c["rejoined"], c["stragglers"], c["straggle_seconds"], st["fallback_steps"], c["partial_steps"], c["poll_resorts"]), flush=True)
timed("warm-up"); timed("steady")
x0 = engs[0].get_positions()
# a spread that asks for a re-sort at the next 64-step poll without outgrowing the shape
for dist in (0.9, 1.1, 1.3):
    B.reset_all(); engs[0].set_positions(_swap_waters_outward(s, x0, n_pairs=1, distance=dist))
    timed("one water %.1f nm out: first 70 steps" % dist, 70)
    for q in range(3):
        timed("   ... 70 more", 70)
    print("      member 0 max_jcount %d" % engs[0].stats()["max_jcount"])
B.reset_all(); engs[0].set_positions(x0)
timed("back to the compact arrangement", 70)
for cycle in range(1):
    engs[0].set_positions(_swap_waters_outward(s, x0))
    timed("member 0 outgrows the shape (cycle %d)" % cycle)
    timed("  ... steps on its own")
    B.reset_all(); engs[0].set_positions(x0)
    timed("next switch: it comes back (cycle %d)" % cycle)
    timed("  ... steady")
B.close()
