import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from blues_amd import systems, integrators, tuning
from blues_amd.engine import NativeEngine
s, v = systems.s23k(mobile_atoms=275, frozen=True)
integ = lambda: integrators.generateNCMCIntegrator(nstepsNC=40, dt=0.004, temperature=300.0, seed=21).to_data(precision=0)
mob = np.nonzero(s.mass > 0)[0]
for skin in (0.20, 0.28, 0.34):
    tuning.reset(); tuning.set(assume_batch=512, skin=skin)
    d = NativeEngine(s, integ()); d.set_velocities(v)
    tuning.reset(); tuning.set(assume_batch=512, skin=skin, k2_dense=0)
    l = NativeEngine(s, integ()); l.set_velocities(v)
    print("skin", skin, "dense", d.stats()["alchemical_kernel"], l.stats()["alchemical_kernel"], "max_jcount", d.stats()["max_jcount"], end=" ")
    for lam in ((0.5, 0.0), (1.0, 0.4)):
        for g in (d, l):
            g.set_global("lambda_sterics", lam[0]); g.set_global("lambda_electrostatics", lam[1])
        ed, el = d.potential_energy(), l.potential_energy()
        fd, fl = d.get_forces()[mob], l.get_forces()[mob]
        print("| dE %.2e dF %.2e" % (abs(ed - el) / abs(el), np.abs(fd - fl).max() / np.abs(fl).max()), end=" ")
    wd, wl = d.run_switch(40, trace=True), l.run_switch(40, trace=True)
    print("| dW %.2e" % (np.abs(wd - wl).max() / np.abs(wl).max()))
    d.close(); l.close()
