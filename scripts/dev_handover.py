"""MD -> NCMC hand-over after an MD leg (reference blues/simulation.py:1028-1037): the frozen NCMC engine receives positions whose
mobile atoms have wandered since its tiles were formed.  Prints what the engine does about it."""
import copy, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blues_amd import build, integrators, systems, tuning
build.build_engine()
from blues_amd.engine import NativeEngine
tuning.set(debug_lists=1, assume_batch=int(sys.argv[1]) if len(sys.argv) > 1 else 16)
s, v = systems.s23k(mobile_atoms=275, frozen=True)
md_sys = copy.copy(systems.s23k(frozen=False)[0]); md_sys.alchemical_atoms = np.zeros(0, np.int32)
g = NativeEngine(s, integrators.generateNCMCIntegrator(nstepsNC=1000, dt=0.004, temperature=300.0, seed=1).to_data(precision=0))
m = NativeEngine(md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=2).to_data(precision=0))
m.set_positions(s.positions); m.set_velocities(v)
g.set_positions(s.positions); g.set_velocities(v)
print("ncmc energy at start", g.potential_energy(), g.stats()["resorts"], flush=True)
for it in range(4):
    m.step(1000)
    x = m.get_positions()
    d = x - s.positions; mob = np.nonzero(s.mass > 0)[0]
    print("iteration %d: max displacement of an NCMC-mobile atom since the start %.3f nm, rms %.3f" % (it, np.sqrt((d[mob] ** 2).sum(1)).max(), np.sqrt((d[mob] ** 2).sum(1).mean())), flush=True)
    g.set_positions(x)
    try:
        print("   ncmc energy", g.potential_energy(), "md energy", m.potential_energy(), g.stats()["resorts"], g.stats()["jcap"], g.stats()["max_jcount"], flush=True)
        tuning.set(debug_lists=0, assume_batch=0)
        fm = NativeEngine(md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=2).to_data(precision=0)); fm.set_positions(x)
        fd = NativeEngine(md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=2).to_data(precision=1)); fd.set_positions(x)
        np.set_printoptions(precision=3, suppress=True, linewidth=200)
        print("   terms ncmc      ", g.energy_terms())
        print("   terms md        ", m.energy_terms(), "audit", m.audit_lists())
        print("   terms md fresh  ", fm.energy_terms(), fm.stats()["nonbonded_kernel"], "audit", fm.audit_lists())
        print("   terms md double ", fd.energy_terms(), flush=True)
        fm.close(); fd.close()
        tuning.set(debug_lists=0, assume_batch=16)
        g.step(50)
        print("   stepped; work", g.get_global("protocol_work"), flush=True)
        g.reset()
    except Exception as e:
        print("   FAILED:", e, flush=True)
        break
