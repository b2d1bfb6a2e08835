"""A batch whose layout shape is outgrown in the middle of a switch (capacity shrunk through BluesTuning.jcap_scale): what the re-plan
for everybody costs, and that the chains come out as they do without one (work within 1e-5 of max|w|).
   python scripts/dev_replan.py [R]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blues_amd import build, integrators, systems, tuning
build.build_engine()
from blues_amd.engine import NativeEngine, NativeBatch
from blues_amd.replicas import replica_seed
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s, v = systems.s23k(mobile_atoms=275, frozen=True)
out = {}
res = np.asarray(s.residue_of_atom)
mob_res = np.unique(res[(s.mass > 0) & (np.arange(s.n_atoms) >= 15)])
frozen_res = np.unique(res[(s.mass == 0)])
for scale in (1.0, 0.62):      # (0.62: member 0's mobile waters are scattered through the box after 200 steps -- no shape holds them)
    tuning.reset(); tuning.set(assume_batch=R)
    engs = []
    for r in range(R):
        g = NativeEngine(s, integrators.generateNCMCIntegrator(nstepsNC=600, dt=0.004, temperature=300.0, seed=replica_seed(77, r)).to_data(precision=0, replica=r)); g.set_velocities(v); engs.append(g)
    B = NativeBatch(engs)
    t = []
    w = None
    for seg in range(6):
        if seg == 2 and scale != 1.0:
            x = engs[0].get_positions()
            partners = np.random.RandomState(3).choice(frozen_res[(frozen_res % 24) != 0], size=len(mob_res), replace=False)
            for a, b in zip(mob_res, partners):
                ia, ib = np.nonzero(res == a)[0], np.nonzero(res == b)[0]
                if len(ia) == 3 and len(ib) == 3:
                    x[ia], x[ib] = x[ib].copy(), x[ia].copy()
            engs[0].set_positions(x)
        t0 = time.perf_counter(); _, w = B.step(100, trace=True); t.append(time.perf_counter() - t0)
    st = engs[0].stats()
    print("jcap_scale %.2f: seconds per 100 steps %s | jcap %d S %d mode %d max_jcount %d resorts %d" % (scale, [round(x, 3) for x in t], st["jcap"], st["tiles_per_list"], st["nonbonded_kernel"], st["max_jcount"], st["resorts"]), flush=True)
    out[scale] = np.array([g.get_global("protocol_work") for g in engs])
    B.close()
    for g in engs: g.close()
d = np.abs(out[1.0][1:] - out[0.62][1:]).max() / np.abs(out[1.0]).max()
print("largest difference of the final protocol work of the OTHER members between the two runs: %.2e of max|w|" % d)
