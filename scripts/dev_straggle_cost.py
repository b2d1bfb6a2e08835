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
    for r in (0, 1):
        e = engs[r].stats(); d = {k: e[k] - last.get((r, k), 0) for k in ("kernel_launches", "list_builds", "atom_prunes", "own_energy_evaluations", "force_passes", "resorts")}
        for k in d: last[(r, k)] = e[k]
        print("      member %d: %s jcap %d tiles/list %d max_jcount %d pruned entries %d" % (r, d, e["jcap"], e["tiles_per_list"], e["max_jcount"], e["pruned_list_entries"]))
    print("%-44s %7.1f ms for %d steps | straggled %d rejoined %d stragglers %d straggle_seconds %.4f fallback %d" % (label, 1e3 * dt, n, c["straggled"], c["rejoined"], c["stragglers"], c["straggle_seconds"], st["fallback_steps"]), flush=True)
timed("warm-up"); timed("steady")
x0 = engs[0].get_positions()
for cycle in range(3):
    engs[0].set_positions(_swap_waters_outward(s, x0))
    timed("member 0 outgrows the shape (cycle %d)" % cycle)
    timed("  ... steps on its own")
    B.reset_all(); engs[0].set_positions(x0)
    timed("next switch: it comes back (cycle %d)" % cycle)
    timed("  ... steady")
B.close()
