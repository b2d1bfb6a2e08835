"""The driver + Move plugins on miniature versions of BASELINE.json's configs[3] (WaterTranslationMove, nothing frozen,
positional restraints) and configs[4] (torsion move on a partially alchemical solute), CPU part: hook behaviour against
what the reference's own tests assert (blues/tests/test_watertranslation.py:95-112)."""
import copy

import numpy as np
import pytest

from blues_amd import integrators, moves, simulation, systems, unit


def water_system(tol_box):
    s, v = tol_box
    w = copy.copy(s)
    w.alchemical_atoms = np.array([15, 16, 17], np.int32)             # the first water is the alchemical one (reference blues/moves.py:889)
    w = systems.restrain_positions(w, np.arange(0, 7), 2092.0)          # "backbone" restraint (examples/water_cuda.yaml:36-38) on the ring carbons
    waters = [[i, i + 1, i + 2] for i in range(15, s.n_atoms, 3)]
    return w, v, waters


def test_water_move_hooks(oracle_backed_context, tol_box):
    w, v, waters = water_system(tol_box)
    integ = integrators.generateNCMCIntegrator(nstepsNC=8, dt=0.002, temperature=300.0, seed=3)
    sim = oracle_backed_context.Simulation(None, w, integ)
    sim.context.setVelocities(unit.Quantity(v, "nanometer/picosecond"))
    mv = moves.WaterTranslationMove(waters, np.arange(15), w.mass[:15], radius=0.9)
    assert mv.atom_indices == [15, 16, 17]
    np.random.seed(4)
    x0 = sim.context.getState(getPositions=True).getPositions(asNumpy=True)._value.copy()
    v0 = sim.context.getState(getVelocities=True).getVelocities(asNumpy=True)._value.copy()
    # protocol work is 0 before any step (test_watertranslation.py:106)
    assert sim.context._integrator.getGlobalVariableByName("protocol_work") == 0
    mv.beforeMove(sim.context)
    x1 = sim.context.getState(getPositions=True).getPositions(asNumpy=True)._value
    v1 = sim.context.getState(getVelocities=True).getVelocities(asNumpy=True)._value
    assert mv.go
    moved = np.nonzero(np.any(x1 != x0, axis=1))[0]
    if len(moved):  # a different water was picked: positions and velocities swapped with the alchemical water
        partner = [i for i in moved if i not in (15, 16, 17)]
        assert len(partner) == 3 and np.array_equal(x1[15:18], x0[partner]) and np.array_equal(x1[partner], x0[15:18])
        assert np.array_equal(v1[15:18], v0[partner]) and np.array_equal(v1[partner], v0[15:18])
    centre = mv._centre(x1)
    assert mv._distance(x1, 15, centre, w.box) <= 0.9
    mv.move(sim.context)
    x2 = sim.context.getState(getPositions=True).getPositions(asNumpy=True)._value
    assert mv._distance(x2, 15, mv._centre(x2), w.box) <= 0.9 + 1e-9
    assert np.allclose(x2[16] - x2[15], x1[16] - x1[15]) and np.allclose(x2[17] - x2[15], x1[17] - x1[15])   # rigid translation
    assert np.array_equal(x2[18:], x1[18:]) and np.array_equal(x2[:15], x1[:15])
    mv.afterMove(sim.context)
    assert sim.context._integrator.getGlobalVariableByName("protocol_work") < 999999
    # out of the sphere -> forced rejection (test_watertranslation.py:108-112)
    x3 = x2.copy(); x3[15:18] += _away(x3, mv, w.box, 1.05)
    sim.context.setPositions(unit.Quantity(x3, "nanometer"))
    mv.afterMove(sim.context)
    assert sim.context._integrator.getGlobalVariableByName("protocol_work") >= 999999
    # no water in range -> the move is disabled for this iteration
    mv2 = moves.WaterTranslationMove(waters, np.arange(15), w.mass[:15], radius=1e-3)
    mv2.beforeMove(sim.context)
    assert mv2.go is False
    assert mv2.move(sim.context) is sim.context


def _away(x, mv, box, dist):
    c = mv._centre(x)
    d = x[15] - c
    d -= box * np.round(d / box)
    u = d / np.linalg.norm(d)
    return u * (dist - np.linalg.norm(d))


def test_torsion_move_geometry(oracle_backed_context, tol_box):
    s, v = tol_box
    integ = integrators.generateNCMCIntegrator(nstepsNC=8, dt=0.002, seed=3)
    sim = oracle_backed_context.Simulation(None, s, integ)
    mv = moves.TorsionRotationMove((1, 0), [7, 8, 9], random_state=5)    # methyl hydrogens about the C2-C1 bond of toluene
    x0 = sim.context.getState(getPositions=True).getPositions(asNumpy=True)._value.copy()
    mv.move(sim.context)
    x1 = sim.context.getState(getPositions=True).getPositions(asNumpy=True)._value
    assert np.array_equal(np.delete(x1, [7, 8, 9], 0), np.delete(x0, [7, 8, 9], 0))
    for h in (7, 8, 9):
        assert np.linalg.norm(x1[h] - x1[0]) == pytest.approx(np.linalg.norm(x0[h] - x0[0]), rel=1e-12)   # C-H length kept
        assert np.linalg.norm(x1[h] - x1[1]) == pytest.approx(np.linalg.norm(x0[h] - x0[1]), rel=1e-12)   # and the angle to the axis
    assert np.linalg.norm(x1[7] - x1[8]) == pytest.approx(np.linalg.norm(x0[7] - x0[8]), rel=1e-12)
    assert 0.0 <= mv.last_angle < 2 * np.pi and np.abs(x1[7] - x0[7]).max() > 1e-3
