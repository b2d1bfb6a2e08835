"""usage (GPU box): python3 scripts/dev_switch_profile.py [R] -- where the HOST time of one switch of bench.py's default workload goes:
cProfile of one_switch after a warm-up switch, and the same switch with the stepping replaced by wall-clock brackets per phase."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np
import torch
import bench
from blues_amd import build, simulation, tuning
build.build_engine()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
with tuning.override(assume_batch=R):
    system, vel, chains = bench.build_chains(0, 0, 1000, "rotmove", R)
drv = simulation.BatchedBLUESSimulation(chains)
st = bench.md_states(chains, system.positions.copy(), vel.copy(), batch=drv._ncmc_batch)
clock = {"sync": 0.0, "switch": 0.0, "decide": 0.0}
bench.one_switch(drv, chains, st, 1000, 0, clock, gather=False)
torch.cuda.synchronize()
# plain stepping, for the difference
b = drv._ncmc_batch
t0 = time.perf_counter(); b.step(1000); torch.cuda.synchronize(); print("1000 plain lockstep steps: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
for k in clock: clock[k] = 0.0
bench.one_switch(drv, chains, st, 1000, 1, clock, gather=False)
torch.cuda.synchronize()
print("one switch:", {k: round(1e3 * v, 1) for k, v in clock.items()})
pr = cProfile.Profile(); pr.enable()
bench.one_switch(drv, chains, st, 1000, 2, clock, gather=False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28); pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
