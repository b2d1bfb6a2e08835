import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blues_amd import systems, integrators
from blues_amd.engine import NativeEngine
s, v = systems.s23k(frozen=False)
md = copy.copy(s); md.alchemical_atoms = np.zeros(0, np.int32)
g = NativeEngine(md, integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=5).to_data(precision=0)); g.set_velocities(v)
g.step(20); st0 = g.stats(); t0 = time.perf_counter(); g.step(200); g.kinetic_energy(); dt = time.perf_counter() - t0; st1 = g.stats()
print("MD leg (23,400 atoms, all mobile): %.1f us/step = %.0f ns/day; launches/step %.2f; rebuilds %d" % (1e6 * dt / 200, 200 * 0.004e-3 / (dt / 86400), (st1["kernel_launches"] - st0["kernel_launches"]) / 200, st1["list_generation"] - st0["list_generation"]))
