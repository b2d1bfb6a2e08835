"""Long all-mobile run in a replica batch: the tiles spread as the liquid diffuses; the engine must re-sort by itself
(resort_hint), keep the temperature, the constraints and never overflow a list."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blues_amd import build, integrators, systems
build.build_engine()
from blues_amd.engine import NativeEngine, NativeBatch
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 30
s, v = systems.s23k(frozen=False)
ndof = 3 * s.n_atoms - len(s.constraint_dist) - 3
engs = []
for r in range(R):
    integ = integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=70 + r)
    g = NativeEngine(s, integ.to_data(precision=0, replica=r)); g.set_velocities(v); engs.append(g)
B = NativeBatch(engs)
t0 = time.perf_counter()
for blk in range(nblocks):
    B.step(200)
    T = [2 * g.kinetic_energy() / (ndof * 0.0083144626) for g in engs]
    st = engs[0].stats()
    print("block %2d  t = %5.1f ps  T = %s  resorts %d  list builds %d  max_jcount %d/%d" % (blk, (blk + 1) * 0.8, " ".join("%.1f" % t for t in T), st["resorts"], st["list_builds"], st["max_jcount"], st["jcap"]), flush=True)
x = engs[0].get_positions(); c = s.constraint_atoms
print("constraint error %.2e" % np.abs(np.linalg.norm(x[c[:, 0]] - x[c[:, 1]], axis=1) / s.constraint_dist - 1).max(), "wall %.1f s" % (time.perf_counter() - t0), B.stats())
