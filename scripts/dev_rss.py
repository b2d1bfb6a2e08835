"""Host memory of the default configuration's set-up, stage by stage: python scripts/dev_rss.py [R]"""
import os, sys, time, gc
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
def rss():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"): return int(line.split()[1]) / 1048576.0
print("start                         %.2f GiB" % rss())
import numpy as np, torch
print("numpy + torch imported        %.2f GiB" % rss())
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
print("HIP initialised               %.2f GiB" % rss())
import bench
from blues_amd import build, tuning, simulation
build.build_engine()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
tuning.set(assume_batch=R // 2)
system, vel, chains = bench.build_chains(0, 0, 1000, "rotmove", R, setup_threads=16)
print("%d chains built             %.2f GiB  (%.2f MiB per chain)" % (R, rss(), 0))
r1 = rss()
drivers = [simulation.BatchedBLUESSimulation(chains[g * R // 2:(g + 1) * R // 2], isolate_failures=True) for g in range(2)]
print("batches                       %.2f GiB" % rss())
st = [bench.md_states(chains[g * R // 2:(g + 1) * R // 2], system.positions.copy(), vel.copy(), batch=drivers[g]._ncmc_batch, driver=drivers[g], decorrelate=250) for g in range(2)]
print("hand-over states              %.2f GiB" % rss())
clock = {"sync": 0.0, "switch": 0.0, "decide": 0.0}
bench.one_switch(drivers[0], chains[:R // 2], st[0], 1000, 0, clock, gather=False)
print("one switch of batch 0         %.2f GiB" % rss())
import resource
print("peak                          %.2f GiB" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0))
