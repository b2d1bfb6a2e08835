import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blues_amd import build, integrators, systems
build.build_engine()
from blues_amd.engine import NativeEngine
s, v = systems.toluene_box()
ndof = 3 * s.n_atoms - len(s.constraint_dist) - 3
integ = integrators.AlchemicalExternalLangevinIntegrator({"lambda_sterics": "1", "lambda_electrostatics": "1"}, splitting="V R O R V", temperature=300.0, timestep=0.004, nsteps_neq=2 ** 30, seed=77)
for prec in (0,):
    g = NativeEngine(s, integ.to_data(precision=prec)); g.set_velocities(v)
    temps, pes = [], []
    for blk in range(int(os.environ.get("NB", "30"))):
        g.step(100); temps.append(2 * g.kinetic_energy() / (ndof * 0.0083144626)); pes.append(g.potential_energy())
    print("precision", prec, "T mean %.1f" % np.mean(temps[5:]), "PE first/last %.0f %.0f" % (pes[0], pes[-1]), g.stats()["list_generation"], g.stats()["max_jcount"], g.stats()["jcap"], flush=True)
    g.close()
