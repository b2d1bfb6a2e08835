"""Where the set-up of R chains goes (bench.py's construction: build_chains -> md_states -> BatchedBLUESSimulation -> first switch warm-up):
wall time per phase, the library's own accounting (blues_debug_setup_seconds) and the top of a cProfile of build_chains.
   python scripts/dev_setupprofile.py [R]"""
import cProfile, os, pstats, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from blues_amd import build, simulation, _lib
build.build_engine()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = _lib.load()
def lib_secs():
    o = (C.c_double * 8)(); lib.blues_debug_setup_seconds(o); return list(o)
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
system, vel, chains = bench.build_chains(0, 0, 1000, "rotmove", R)
pr.disable()
t1 = time.perf_counter()
a = lib_secs()
states = bench.md_states(chains, system.positions.copy(), vel.copy())
t2 = time.perf_counter()
b = lib_secs()
drv = simulation.BatchedBLUESSimulation(chains)
t3 = time.perf_counter()
clock = {"sync": 0.0, "switch": 0.0, "decide": 0.0}
bench.one_switch(drv, chains, states, 1000, 0, clock, gather=False)
t4 = time.perf_counter()
c = lib_secs()
print("R=%d  build_chains %.2f s | md_states %.2f s | batch %.2f s | first 1000-step switch %.2f s" % (R, t1 - t0, t2 - t1, t3 - t2, t4 - t3))
names = ["engine_create", "sort_and_tile", " (image+uploads)", "hipMalloc", "memset", "H2D copies", "allocations", "streams+events"]
for nm, x, y, z in zip(names, a, b, c):
    print("  %-18s after build_chains %8.3f  after md_states %8.3f  after first switch %8.3f" % (nm, x, y, z))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
