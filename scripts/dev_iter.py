import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blues_amd import systems, integrators, unit
from blues_amd.engine import NativeEngine
s, v = systems.toluene_box()
md_sys = copy.copy(s); md_sys.alchemical_atoms = np.zeros(0, np.int32)
md = NativeEngine(md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=5).to_data())
md.set_velocities(v)
nc = NativeEngine(s, integrators.generateNCMCIntegrator(nstepsNC=10, dt=0.004, temperature=300.0, seed=4).to_data())
nc.set_velocities(v)
x = md.get_positions(); vv = md.get_velocities()
nc.set_positions(x); nc.set_velocities(vv)
c = s.constraint_atoms
print("md constraint err", np.abs(np.linalg.norm(x[c[:,0]]-x[c[:,1]],axis=1)/s.constraint_dist-1).max())
try:
    nc.step(5); print("ncmc 5 steps ok")
except Exception as e: print("ncmc fail:", e)
try:
    md.step(10); print("md 10 ok")
except Exception as e: print("md fail:", e)
x = md.get_positions()
print("md constraint err", np.abs(np.linalg.norm(x[c[:,0]]-x[c[:,1]],axis=1)/s.constraint_dist-1).max())
