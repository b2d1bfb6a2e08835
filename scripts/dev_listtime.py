import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blues_amd import build, integrators, systems
build.build_engine()
from blues_amd.engine import NativeEngine
for frozen in (True, False):
    system, vel = systems.s23k(mobile_atoms=275, frozen=frozen)
    g = NativeEngine(system, integrators.generateNCMCIntegrator(nstepsNC=100, dt=0.004, temperature=300.0, seed=1).to_data(precision=0))
    g.set_velocities(vel); g.step(5)
    print("frozen" if frozen else "all-mobile", "tiles", g.stats()["i_tiles"], "list build %.1f us" % g.time_list_build(50), "K1 %.1f us" % g.time_nonbonded(50), flush=True)
