"""Where a chain's set-up goes on the host: cProfile of bench.build_chains for R chains on one thread: python scripts/dev_setup_profile.py [R]"""
import cProfile, pstats, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench
from blues_amd import build, tuning
build.build_engine()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tuning.set(assume_batch=1024)
bench.build_chains(0, 0, 1000, "rotmove", 8, setup_threads=1)      # warm: library, tables
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
system, vel, chains = bench.build_chains(0, 0, 1000, "rotmove", R, setup_threads=1)
pr.disable()
print("%d chains on one thread: %.3f s = %.2f ms per chain" % (R, time.perf_counter() - t0, 1e3 * (time.perf_counter() - t0) / R))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
t0 = time.perf_counter()
system, vel, chains2 = bench.build_chains(0, 0, 1000, "rotmove", 1024, setup_threads=16)
print("1024 chains on 16 threads: %.3f s" % (time.perf_counter() - t0))
from blues_amd import _lib
import ctypes
try:
    parts = (ctypes.c_double * 8)(); _lib.load().blues_debug_setup_seconds(parts); print("native setup seconds by part:", list(parts))
except Exception as e:
    print("no native breakdown:", e)
