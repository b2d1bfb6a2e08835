"""Per-call host cost of the per-chain boundary operations (engine inside a replica batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blues_amd import build, integrators, systems
build.build_engine()
from blues_amd.engine import NativeEngine, NativeBatch
system, vel = systems.s23k(mobile_atoms=275, frozen=True)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 32
engs = []
for r in range(R):
    g = NativeEngine(system, integrators.generateNCMCIntegrator(nstepsNC=1000, dt=0.004, temperature=300.0, seed=r).to_data(precision=0, replica=r))
    g.set_velocities(vel); engs.append(g)
B = NativeBatch(engs); B.step(20)
g = engs[3]
x = g.get_positions(); v = g.get_velocities()
def t(name, fn, n=100):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    print("%-34s %8.1f us" % (name, 1e6 * (time.perf_counter() - t0) / n), flush=True)
t("kinetic_energy", g.kinetic_energy)
t("potential_energy (cached)", g.potential_energy)
def pe_fresh(): g.set_global("lambda_sterics", 0.5 if g.get_global("lambda_sterics") != 0.5 else 0.4); g.potential_energy()
t("potential_energy (evaluated)", pe_fresh)
t("get_global protocol_work", lambda: g.get_global("protocol_work"))
t("get_global lambda", lambda: g.get_global("lambda"))
snaps = []
def cap(): s = g.snapshot(True, True); snaps.append(s); (snaps.pop(0).release() if len(snaps) > 2 else None)
t("snapshot capture x+v", cap)
s0 = g.snapshot(True, True)
t("set_positions_from_snapshot", lambda: g.set_positions_from_snapshot(s0))
t("set_velocities_from_snapshot", lambda: g.set_velocities_from_snapshot(s0))
t("snapshot read positions", lambda: (s0._host.clear(), s0.read(1)))
t("set_positions (host)", lambda: g.set_positions(x))
t("set_velocities (host)", lambda: g.set_velocities(v))
t("get_positions", g.get_positions)
t("set_velocities_to_temperature", lambda: g.set_velocities_to_temperature(300.0, 5))
t("reset", g.reset)
t("get_box", g.get_box)
t("batch.step(1)", lambda: B.step(1), 50)
t("batch.step(10)", lambda: B.step(10), 20)
