#!/usr/bin/env python3
"""Developer probe: us/step of the benchmark switch under different decomposition/skin settings."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, %r)
from blues_amd import systems, integrators
from blues_amd.engine import NativeEngine
s, v = systems.s23k(mobile_atoms=275, frozen=(os.environ.get("FULL","0")!="1"))
n = int(os.environ.get("NST","500"))
integ = integrators.generateNCMCIntegrator(nstepsNC=n, dt=0.004, temperature=300.0, seed=3)
g = NativeEngine(s, integ.to_data(precision=0)); g.set_velocities(v)
g.run_switch(20)
st0 = g.stats(); t0 = time.perf_counter(); g.run_switch(n - 40); dt = time.perf_counter() - t0; st1 = g.stats()
print("%%7.1f us/step  rebuilds %%3d  launches/step %%.2f  %%s  K1 %%.1f us" %% (1e6 * dt / (n - 40), st1["list_generation"] - st0["list_generation"], (st1["kernel_launches"] - st0["kernel_launches"]) / (n - 40), {k: st1[k] for k in ("i_tiles","jcap","npart","seg_len","wpb")}, g.time_nonbonded(20)))
''' % ROOT
for env in sys.argv[1:] or ["", "BLUES_SEG=4", "BLUES_SEG=16", "BLUES_SKIN=0.2", "BLUES_SKIN=0.2 BLUES_SEG=4", "BLUES_SKIN=0.3"]:
    e = dict(os.environ); e.update(dict(kv.split("=") for kv in env.split()))
    out = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
    print("%-28s %s" % (env or "(default)", (out.stdout.strip().split("\n") or [""])[-1] or out.stderr[-300:]))
