import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blues_amd import build
build.LIB_PATH = os.path.join(build.CSRC, "libblues_hip_stamp.so")
import blues_amd._lib as L; L.LIB_PATH = build.LIB_PATH
from blues_amd import systems, integrators
from blues_amd.engine import NativeEngine
s, v = systems.s23k(mobile_atoms=275)
integ = integrators.generateNCMCIntegrator(nstepsNC=100, dt=0.004, temperature=300.0, seed=3)
g = NativeEngine(s, integ.to_data(precision=0)); g.set_velocities(v)
g.run_switch(50)
print("ops: V0 H01 END CM H12 V2 R O R | tail(write-back)")
g.close()
