#!/usr/bin/env python3
"""Developer probe: wall time per NCMC step for different splittings (isolates the cost of each op)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blues_amd import systems, integrators
from blues_amd.engine import NativeEngine

s, v = systems.s23k(mobile_atoms=275)
n = 400
for split in ("H V R O R V H", "V R O R V", "V R R V", "V O V", "V V", "R R", "O", "R", "V"):
    integ = integrators.AlchemicalExternalLangevinIntegrator(integrators.DEFAULT_ALCHEMICAL_FUNCTIONS, splitting=split, temperature=300.0,
                                                              timestep=0.001, nsteps_neq=n, seed=3)
    g = NativeEngine(s, integ.to_data(precision=0)); g.set_velocities(v)
    g.run_switch(20)
    st0 = g.stats(); t0 = time.perf_counter(); g.run_switch(n - 40); dt = time.perf_counter() - t0; st1 = g.stats()
    print("%-16s %7.1f us/step  launches/step %.2f  passes/step %.2f" % (split, 1e6 * dt / (n - 40), (st1["kernel_launches"] - st0["kernel_launches"]) / (n - 40), (st1["force_passes"] - st0["force_passes"]) / (n - 40)))
    g.close()
