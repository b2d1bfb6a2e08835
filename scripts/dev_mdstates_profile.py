import cProfile, os, pstats, sys, time
sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import numpy as np
import bench
from blues_amd import build, simulation, tuning
build.build_engine()
R = 128
with tuning.override(assume_batch=R):
    system, vel, chains = bench.build_chains(0, 0, 1000, "rotmove", R)
drv = simulation.BatchedBLUESSimulation(chains)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
st = bench.md_states(chains, system.positions.copy(), vel.copy(), batch=drv._ncmc_batch)
print("md_states", time.perf_counter() - t0)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
