"""Phase stamps of the batched per-atom-list force kernel under load (1024 chains): python scripts/dev_k1stamps.py [R]
   needs the -DBLUES_STAMP build: hipcc ... -DBLUES_STAMP -o blues_amd/csrc/libblues_hip_stamp.so (scripts/build_stamp.sh)"""
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
    g = NativeEngine(s, integrators.generateNCMCIntegrator(nstepsNC=400, dt=0.004, temperature=300.0, seed=replica_seed(1234, r)).to_data(precision=0, replica=r)); g.set_velocities(v); engs.append(g)
B = NativeBatch(engs)
B.step(100)
for k in range(3):
    print("K1 %.1f us" % B.time_nonbonded(20), flush=True)
    B.step(7)
B.close()
