"""Host-side profile of one FULL iteration (bench.py --md-steps): where the wall time of the NCMC leg and of the boundary goes.
   python scripts/dev_full_profile.py [R] [md_steps]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from blues_amd import build, simulation, tuning, unit
build.build_engine()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
md = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
with tuning.override(assume_batch=R):
    system, vel, chains = bench.build_chains(0, 0, 1000, "rotmove", R, md_steps=md)
drv = simulation.BatchedBLUESSimulation(chains)
for c in chains:
    c._md_sim.context.setPositions(unit.Quantity(system.positions, "nanometer")); c._md_sim.context.setVelocities(unit.Quantity(vel, "nanometer/picosecond"))
clock = {"sync": 0.0, "switch": 0.0, "decide": 0.0}
for it in range(2):
    bench.one_iteration(drv, chains, 1000, md, it, clock)
clock = {"sync": 0.0, "switch": 0.0, "decide": 0.0}
pr = cProfile.Profile(); pr.enable()
bench.one_iteration(drv, chains, 1000, md, 2, clock)
pr.disable()
print(clock, chains[0]._ncmc_sim.context._engine.stats()["nonbonded_kernel"], drv._ncmc_batch.stats())
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
