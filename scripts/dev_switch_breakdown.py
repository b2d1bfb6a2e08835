import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import torch
system, vel, sim, blues = bench.build_replica(0, 0, 1000, "rotmove")
x0 = system.positions.copy(); v0 = vel.copy()
bench.one_switch(blues, sim, x0, v0, 1000, 0)
def T(label, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); print("%-28s %8.2f ms" % (label, 1e3 * (time.perf_counter() - t))); return r
for it in range(2):
    print("--- switch", it)
    T("setPositions(x0)", lambda: sim.context.setPositions(x0))
    T("setVelocities(v0)", lambda: sim.context.setVelocities(v0))
    T("_syncStatesMDtoNCMC", lambda: blues._syncStatesMDtoNCMC())
    T("_stepNCMC", lambda: blues._stepNCMC(1000, 500))
    T("_acceptRejectMove", lambda: blues._acceptRejectMove())
    T("_resetSimulations", lambda: blues._resetSimulations(300.0))
e = sim.context._engine
T("potential_energy", lambda: e.potential_energy())
T("kinetic_energy", lambda: e.kinetic_energy())
T("get_positions", lambda: e.get_positions())
T("run_switch(500)", lambda: (e.reset(), e.run_switch(500)))
