"""List validity between rebuilds: after many steps, forces evaluated with the lists as they stand (not rebuilt) against the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blues_amd import build, integrators, systems
build.build_engine()
from blues_amd.engine import NativeEngine
from oracle import oracle
s, v = systems.toluene_box()
integ = integrators.AlchemicalExternalLangevinIntegrator({"lambda_sterics": "1", "lambda_electrostatics": "1"}, splitting="V R O R V", temperature=300.0, timestep=0.004, nsteps_neq=2 ** 30, seed=77)
g = NativeEngine(s, integ.to_data(precision=0)); g.set_velocities(v)
worst = 0.0
o = oracle.Oracle(s, integ.to_data())
for blk in range(40):
    g.step(37)
    f = g.get_forces(); x = g.get_positions()
    o.set_positions(x); fo = o.energy_forces(1.0, 1.0)[1]
    err = np.abs(f - fo).max() / np.abs(fo).max()
    if err > 1e-5 and worst <= 1e-5:
        bad = np.argsort(-np.abs(f - fo).max(1))[:5]
        g2 = NativeEngine(s, integ.to_data(precision=0)); g2.set_positions(x); f2 = g2.get_forces(); g2.close()
        print("  first bad check blk", blk, "err %.2e" % err, "atoms", bad.tolist(), "abs", np.abs(f - fo).max(1)[bad].round(3).tolist(), "fresh-engine err %.2e" % (np.abs(f2 - fo).max() / np.abs(fo).max()), "lists", g.stats()["list_generation"])
    worst = max(worst, err)
print("mode env", os.environ.get("BLUES_FUSE"), os.environ.get("BLUES_K1_MODE"), "worst relative force error over 40 checks: %.2e" % worst, g.stats()["list_generation"])
