import copy, sys, numpy as np
sys.path.insert(0, '/root/repo')
from blues_amd import build, integrators, systems, tuning
build.build_engine()
from blues_amd.engine import NativeEngine
s, v = systems.toluene_box()
res = np.asarray(s.residue_of_atom)
keep = np.nonzero((res % 3 == 0) | (np.arange(s.n_atoms) < 15))[0]
fz = systems.freeze_except(copy.copy(s), keep)
v = v.copy(); v[fz.mass == 0.0] = 0.0
for mode in (-1, 3):
    tuning.reset(); tuning.set(k1_mode=mode)
    g = NativeEngine(fz, integrators.generateNCMCIntegrator(nstepsNC=40, dt=0.002, temperature=300.0, seed=11).to_data(precision=0))
    g.set_velocities(v)
    try:
        for seg in range(8):
            w = g.run_switch(5, trace=True)
            print(mode, seg, w[-1], g.stats()["nonbonded_kernel"], g.audit_lists() if mode == 3 else "", g.stats()["list_builds"], g.stats()["atom_prunes"], flush=True)
    except Exception as e:
        print(mode, "FAILED", e)
        f = g.get_forces(); print(np.abs(f).max(), np.argmax(np.abs(f).max(1)))
