"""Diagnostic for tests/test_gpu_pme.py::test_differential_correction_equals_the_four_energy_form[True]: where do the two forms part?"""
import os, sys, copy
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from blues_amd import build, integrators, systems, tuning, moves, simulation, unit
build.build_engine()
from blues_amd.context import Simulation
s, v = systems.toluene_box()
s = systems.with_reciprocal_space(s)
md_sys = copy.copy(s); md_sys.alchemical_atoms = np.zeros(0, np.int32)
lig = np.arange(15)
R, nsteps, nmd, nIter = 3, 10, 6, 2
tuning.set(assume_batch=R)
def chains(differential):
    out = []
    for r in range(R):
        sim = Simulation(None, s, integrators.generateNCMCIntegrator(nstepsNC=nsteps, dt=0.002, temperature=300.0, seed=700 + r), precision="double", replica=r)
        md = Simulation(None, md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.002, seed=800 + r), precision="double", replica=r)
        alch = Simulation(None, md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.002, seed=900 + r), precision="double", replica=r)
        md.context.setPositions(unit.Quantity(s.positions, "nanometer")); md.context.setVelocities(unit.Quantity(v * (1.0 + 0.03 * r), "nanometer/picosecond"))
        mover = moves.MoveEngine(moves.RandomLigandRotationMove(lig, s.mass[lig], random_state=90 + r))
        out.append(simulation.BLUESSimulation(simulation.SimulationSet(sim, md=md, alch=alch), {"nstepsNC": nsteps, "moveStep": nsteps // 2, "nIter": nIter, "nstepsMD": nmd},
                                              mover, rng=np.random.RandomState(4000 + r), differential_correction=differential))
    return out
res = {}
for differential in (False, None):
    cs = chains(differential)
    B = simulation.BatchedBLUESSimulation(cs, batched_boundary=True)
    recs = []
    def note(N, last):
        recs.append(([dict(l) for l in last], [c._md_sim.context._engine.get_positions().copy() for c in cs], [c._ncmc_sim.context._engine.get_positions().copy() for c in cs]))
    B.run(nIter=nIter, on_iteration=note)
    res[differential] = recs
    B.close()
for N in range(nIter):
    for r in range(R):
        a, b = res[None][N][0][r], res[False][N][0][r]
        print("iteration %d chain %d: correction %.12f / %.12f (diff %.2e)  work %.12f / %.12f (diff %.2e) accept %s/%s | md x diff %.2e ncmc x diff %.2e"
              % (N, r, a["correction"], b["correction"], a["correction"] - b["correction"], a["protocol_work"], b["protocol_work"], a["protocol_work"] - b["protocol_work"], a["accept"], b["accept"],
                 np.abs(res[None][N][1][r] - res[False][N][1][r]).max(), np.abs(res[None][N][2][r] - res[False][N][2][r]).max()))
