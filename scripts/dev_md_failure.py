"""Hunting a rare MD-leg failure of the full iteration (GPU box): bench.py's full iteration (md + alch + ncmc Simulations per chain, batched
boundary) for many iterations; when a chain's MD leg raises, everything that says WHAT happened to that chain is printed: the iteration, its
last NCMC decision, the MD engine's counters, how far its atoms are from where the leg started, the closest pair of atoms, its kinetic energy.
   python scripts/dev_md_failure.py [--R 64] [--iters 60] [--seed 0]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from blues_amd import build, simulation, tuning
build.build_engine()

ap = argparse.ArgumentParser()
ap.add_argument("--R", type=int, default=64)
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
np.random.seed(a.seed)
nsteps = 1000
with tuning.override(assume_batch=a.R):
    system, vel, chains = bench.build_chains(0, 0, nsteps, "rotmove", a.R, md_steps=1000, with_alch=True)
driver = simulation.BatchedBLUESSimulation(chains)
from blues_amd import unit
x0 = system.positions.copy()
for c in chains:   # (as bench.py --md-steps: the MD contexts own the state)
    c._md_sim.context.setPositions(unit.Quantity(x0, "nanometer")); c._md_sim.context.setVelocities(unit.Quantity(vel.copy(), "nanometer/picosecond"))


def min_pair(x, box):
    """closest pair of atoms (cell list over 0.3 nm cells)"""
    L = np.asarray(box)
    f = (x / L) % 1.0
    nc = np.maximum(1, (L / 0.3).astype(int))
    cell = (f * nc).astype(int) % nc
    key = (cell[:, 0] * nc[1] + cell[:, 1]) * nc[2] + cell[:, 2]
    order = np.argsort(key); best = (1e9, -1, -1)
    start = np.searchsorted(key[order], np.arange(nc.prod() + 1))
    for i in range(len(x)):
        c = cell[i]
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    k = (((c[0] + dx) % nc[0]) * nc[1] + (c[1] + dy) % nc[1]) * nc[2] + (c[2] + dz) % nc[2]
                    js = order[start[k]:start[k + 1]]
                    js = js[js > i]
                    if len(js) == 0: continue
                    d = x[js] - x[i]; d -= L * np.round(d / L)
                    r = np.sqrt((d * d).sum(1)); q = r.argmin()
                    if r[q] < best[0]: best = (float(r[q]), i, int(js[q]))
    return best


clock = {"sync": 0.0, "switch": 0.0, "decide": 0.0, "md": 0.0}
orig_advance = driver._advance
info = {"it": -1}


def watched(batch, sims, wanted):
    errors = orig_advance(batch, sims, wanted)
    if errors and batch is driver._md_batch:
        for r, e in errors.items():
            c = chains[r]
            eng = c._md_sim.context._engine
            print("== MD leg of chain %d failed in iteration %d: %s" % (r, info["it"], e), flush=True)
            print("   its last NCMC decision:", {k: c.last.get(k) for k in ("accept", "log_accept", "protocol_work", "correction")})
            try: print("   MD engine:", eng.stats())
            except Exception as ex: print("   stats failed:", ex)
            try:
                x = eng.get_positions(); v = eng.get_velocities()
                bad = ~np.isfinite(x).all(1)
                print("   positions: %d atoms not finite; |v| max %.2f nm/ps (atom %d), 99.9th percentile %.2f" % (bad.sum(), np.nanmax(np.abs(v)), int(np.nanargmax(np.abs(v).max(1))), np.nanpercentile(np.abs(v), 99.9)))
                st = c.stateTable["md"]["state0"] if "md" in c.stateTable else None
                if st is not None:
                    x_start = np.asarray(st["positions"]._value if hasattr(st["positions"], "_value") else st["positions"])
                    d = x - x_start; d -= np.asarray(eng.get_box()) * np.round(d / np.asarray(eng.get_box()))
                    r2 = np.sqrt((d * d).sum(1))
                    print("   displacement since the leg's start: rms %.3f max %.3f nm (atom %d)" % (np.sqrt(np.nanmean(r2 ** 2)), np.nanmax(r2), int(np.nanargmax(r2))))
                if not bad.any():
                    print("   closest pair of atoms now: %.4f nm (atoms %d, %d)" % min_pair(x, eng.get_box()))
            except Exception as ex:
                print("   reading the state failed:", ex)
            nce = c._ncmc_sim.context._engine
            try: print("   its NCMC engine:", {k: nce.stats()[k] for k in ("nonbonded_kernel", "resorts", "list_builds")})
            except Exception as ex: print("   NCMC stats failed:", ex)
        print("   (other chains: %d of %d fine)" % (len(sims) - len(errors), len(sims)), flush=True)
        sys.exit(3)
    return errors


driver._advance = watched
t0 = time.perf_counter()
for it in range(a.iters):
    info["it"] = it
    bench.one_iteration(driver, chains, nsteps, 1000, it, clock)
    if it % 10 == 9:
        acc = np.mean([c.last["accept"] for c in chains])
        print("iteration %d done (%.1f s); accepted in this one: %.2f" % (it + 1, time.perf_counter() - t0, acc), flush=True)
print("no failure in %d iterations of %d chains" % (a.iters, a.R))
