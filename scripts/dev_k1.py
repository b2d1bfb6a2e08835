import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, %r)
import numpy as np
from blues_amd import systems, integrators
from blues_amd.engine import NativeEngine
s, v = systems.s23k(frozen=False)
integ = integrators.generateNCMCIntegrator(nstepsNC=100, dt=0.004, temperature=300.0, seed=3)
g = NativeEngine(s, integ.to_data(precision=0)); g.set_velocities(v)
e = g.potential_energy(); f = g.get_forces()
t = g.time_nonbonded(30)
g.run_switch(10); t0 = time.perf_counter(); g.run_switch(60); dt = time.perf_counter() - t0
print("K1 %%7.1f us  step %%7.1f us  E %%.6f  |F| %%.6f  %%s" %% (t, 1e6 * dt / 60, e, np.linalg.norm(f), {k: g.stats()[k] for k in ("npart", "seg_len", "wpb")}))
''' % ROOT
for env in sys.argv[1:]:
    e = dict(os.environ); e.update(dict(kv.split("=") for kv in env.split()))
    out = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
    print("%-26s %s" % (env or "(default)", (out.stdout.strip().split("\n") or [""])[-1] or out.stderr[-400:]))
