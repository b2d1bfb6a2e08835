"""Phase stamps of the batched per-atom-list force kernel and the dense alchemical kernel under load (1024 chains): python scripts/dev_k1stamps.py [R]
   needs the -DBLUES_STAMP build: bash scripts/build_stamp.sh"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BLUES_LIB_PATH"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "blues_amd", "csrc", "libblues_hip_stamp.so")
from blues_amd import integrators, systems
from blues_amd.engine import NativeEngine, NativeBatch
from blues_amd.replicas import replica_seed
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
s, v = systems.s23k(mobile_atoms=275, frozen=True)
engs = []
for r in range(R):
    g = NativeEngine(s, integrators.generateNCMCIntegrator(nstepsNC=1000, dt=0.004, temperature=300.0, seed=replica_seed(1234, r)).to_data(precision=0, replica=r)); g.set_velocities(v); engs.append(g)
B = NativeBatch(engs)
for first in (100, 250, 150):      # (stamps while electrostatics move, while sterics move, while nothing moves in the middle ... each after some steps of that regime)
    B.step(first)
    print("after %d more steps: K1 %.1f us" % (first, B.time_nonbonded(20)), flush=True)
B.close()
