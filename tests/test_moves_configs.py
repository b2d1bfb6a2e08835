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


def test_sidechain_move_from_the_bond_graph(tol_box):
    """SideChainMove without OpenEye (reference blues/moves.py:418-844): rotatable heavy bonds and rotating atoms from the bonded
    graph.  On toluene (residue 0) the only rotor candidate ring-methyl bond C1-C7 is terminal on the methyl side for heavy atoms
    (no heavy neighbour beyond C7), so OEBond::IsRotor() and this predicate agree there is no heavy rotor; a butane-like chain has one."""
    from blues_amd import moves
    from blues_amd._abi import SystemData
    s, _ = tol_box
    with pytest.raises(Exception):
        m = moves.SideChainMove(s, [0]); m.chooseBondandTheta()          # no heavy rotor in toluene: nothing to choose from
    # n-butane C0-C1-C2-C3 with hydrogens; a ring (cyclopropane C10-C11-C12) whose bonds must not be rotors; one side chain on a backbone
    names = ["C1", "C2", "C3", "C4"] + ["H%d" % k for k in range(10)]
    bonds = [(0, 1), (1, 2), (2, 3)] + [(0, 4), (0, 5), (0, 6), (1, 7), (1, 8), (2, 9), (2, 10), (3, 11), (3, 12), (3, 13)]
    n = 14
    rng = np.random.RandomState(0)
    x = rng.normal(size=(n, 3)) * 0.1
    x[:4] = [[0, 0, 0], [0.15, 0, 0], [0.2, 0.14, 0], [0.35, 0.14, 0.02]]
    sysd = SystemData(box=np.array([3.0, 3.0, 3.0]), mass=np.array([12.0] * 4 + [1.0] * 10), charge=np.zeros(n), sigma=np.full(n, 0.3), epsilon=np.zeros(n),
                      bond_atoms=np.array(bonds, np.int32), bond_params=np.tile([0.15, 1000.0], (len(bonds), 1)), positions=x,
                      residue_of_atom=np.zeros(n, np.int32), names=names)
    m = moves.SideChainMove(sysd, [0], random_state=4)
    assert list(m.rot_bonds) == [(1, 2)]                                   # C1-C2 and C3-C4 are terminal, C2-C3 is the heavy rotor
    atoms = m.rot_atoms[0][(1, 2)]
    assert atoms[:2] == [1, 2] and set(atoms) == set(range(n))            # the reference's walk starts from both axis atoms (no backbone here)
    R = m.rotation_matrix([0, 0, 1.0], 0.3)
    assert np.allclose(R @ R.T, np.eye(3)) and np.isclose(np.linalg.det(R), 1.0) and np.allclose(R @ [0, 0, 1.0], [0, 0, 1.0])
    theta, target, res, bond = m.chooseBondandTheta()
    assert 0 <= theta < 2 * np.pi and res == 0 and bond == (1, 2) and m.alchemical_atoms == list(range(n))
    # a protein-like residue: backbone N, CA, C, O + side chain CB-CG-CD: rotors CA-CB and CB-CG; the walk never crosses into the backbone
    names2 = ["N", "CA", "C", "O", "CB", "CG", "CD", "HB", "HG"]
    bonds2 = [(0, 1), (1, 2), (2, 3), (1, 4), (4, 5), (5, 6), (4, 7), (5, 8)]
    s2 = SystemData(box=np.array([3.0, 3.0, 3.0]), mass=np.array([14.0, 12, 12, 16, 12, 12, 12, 1, 1]), charge=np.zeros(9), sigma=np.full(9, 0.3), epsilon=np.zeros(9),
                    bond_atoms=np.array(bonds2, np.int32), bond_params=np.tile([0.15, 1000.0], (len(bonds2), 1)), positions=rng.normal(size=(9, 3)),
                    residue_of_atom=np.zeros(9, np.int32), names=names2)
    m2 = moves.SideChainMove(s2, [0])
    assert sorted(m2.rot_bonds) == [(1, 4), (4, 5)] and m2.backbone_atoms == [0, 1, 2, 3]
    assert m2.rot_atoms[0][(1, 4)][:2] == [1, 4] and set(m2.rot_atoms[0][(1, 4)]) == {1, 4, 5, 6, 7, 8}    # CA is the axis, N / C / O stay
    assert set(m2.rot_atoms[0][(4, 5)]) == {4, 5, 6, 7, 8}
