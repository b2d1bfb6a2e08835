import os, sys
os.environ.setdefault("BLUES_TUNING", "assume_batch=256")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blues_amd import build
build.LIB_PATH = os.path.join(build.CSRC, "libblues_hip_stamp.so")
import blues_amd._lib as L; L.LIB_PATH = build.LIB_PATH
from blues_amd import systems, integrators
from blues_amd.engine import NativeEngine
s, v = systems.s23k(mobile_atoms=275, frozen=True)
r = systems.with_reciprocal_space(s)
g = NativeEngine(r, integrators.generateNCMCIntegrator(nstepsNC=100, dt=0.004, temperature=300.0, seed=3).to_data(precision=0)); g.set_velocities(v)
g.step(3)
f = g.get_forces()
print("nonbonded %.1f us" % g.time_nonbonded(1), flush=True)
