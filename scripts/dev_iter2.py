import sys, os, copy, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blues_amd import systems, integrators, unit, moves, simulation
from blues_amd.context import Simulation
s, v = systems.toluene_box()
md_sys = copy.copy(s); md_sys.alchemical_atoms = np.zeros(0, np.int32)
ncmc = Simulation(None, s, integrators.generateNCMCIntegrator(nstepsNC=10, dt=0.004, temperature=300.0, seed=4))
md = Simulation(None, md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=5))
for sim in (ncmc, md):
    sim.context.setVelocities(unit.Quantity(v, "nanometer/picosecond"))
lig = np.arange(15)
mv = moves.RandomLigandRotationMove(lig, s.mass[lig], random_state=2)
e = ncmc.context._engine
c = s.constraint_atoms
def cerr(x): return np.abs(np.linalg.norm(x[c[:,0]]-x[c[:,1]],axis=1)/s.constraint_dist-1).max()
for it in range(2):
    st = md.context.getState(getPositions=True, getVelocities=True)
    ncmc.context.setPositions(st.getPositions(asNumpy=True)); ncmc.context.setVelocities(st.getVelocities(asNumpy=True))
    print(it, "synced; cerr", cerr(e.get_positions()), "PE", e.potential_energy())
    for k in range(10):
        if k == 5:
            mv.move(ncmc.context); print("  moved; cerr", cerr(e.get_positions()))
        try:
            ncmc.step(1)
        except Exception as ex:
            print("  step", k, "FAILED", ex); x = e.get_positions(); print("  cerr", cerr(x), "nan", np.isnan(x).any()); sys.exit(0)
    print("  work", e.get_global("protocol_work"), "cerr", cerr(e.get_positions()))
    ncmc.currentStep = 0; ncmc.integrator.reset()
    md.context.setVelocitiesToTemperature(300.0, 5 + it)
    md.step(10)
    print("  md done cerr", cerr(md.context._engine.get_positions()))
