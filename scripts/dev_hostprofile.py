"""cProfile of one batched switch through the BLUES driver mirror (where does the host time go?)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from blues_amd import simulation
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
system, vel, chains = bench.build_chains(0, 0, n, "rotmove", R)
x0 = system.positions.copy(); v0 = vel.copy()
drv = simulation.BatchedBLUESSimulation(chains)
states = bench.md_states(chains, x0, v0)
clock = {"sync": 0.0, "switch": 0.0, "decide": 0.0}
bench.one_switch(drv, chains, states, n, 0, clock)
clock = {"sync": 0.0, "switch": 0.0, "decide": 0.0}
pr = cProfile.Profile(); pr.enable()
bench.one_switch(drv, chains, states, n, 1, clock)
pr.disable()
print(clock)
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
st = pstats.Stats(pr)
st.sort_stats("cumtime").print_stats(45)
st.print_callers("get_global")
st.print_callers("getState")
