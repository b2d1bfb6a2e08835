"""Phase stamps (s_memtime, 100 MHz) of the list build and the per-atom-list force kernel on a lone S23k engine in the
large-batch decomposition (BLUES_FUSE=0).  Uses the -DBLUES_STAMP build blues_amd/csrc/libblues_hip_stamp.so."""
import os, sys
os.environ.setdefault("BLUES_TUNING", "assume_batch=256")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blues_amd import build
build.LIB_PATH = os.path.join(build.CSRC, "libblues_hip_stamp.so")
import blues_amd._lib as L; L.LIB_PATH = build.LIB_PATH
from blues_amd import systems, integrators
from blues_amd.engine import NativeEngine
for frozen in (True, False):
    s, v = systems.s23k(mobile_atoms=275, frozen=frozen)
    g = NativeEngine(s, integrators.generateNCMCIntegrator(nstepsNC=100, dt=0.004, temperature=300.0, seed=3).to_data(precision=0)); g.set_velocities(v)
    g.step(5)
    st = g.stats()
    print("frozen" if frozen else "all-mobile", "tiles", st["i_tiles"], "jcap", st["jcap"], "max_jcount", st["max_jcount"], flush=True)
    print("  list build %.1f us" % g.time_list_build(20), flush=True)
    print("  nonbonded  %.1f us" % g.time_nonbonded(20), flush=True)
    g.close()
