"""GPU suite: reciprocal space (BLUES_NB_PME: smooth-PME mesh, self term, excluded-pair corrections, dispersion correction;
SURVEY.md 8f.2) of the HIP engine against the oracle -- energies, forces, the split into a static frozen part and a per-step
mobile part, stepping, batches, and the alchemical correction that stops being identically zero."""
import copy

import numpy as np
import pytest

from blues_amd import integrators, systems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine():
    from blues_amd import build
    build.build_engine()
    from blues_amd.engine import NativeEngine
    return NativeEngine


def _integ(n=20, seed=7, **kw):
    return integrators.generateNCMCIntegrator(nstepsNC=n, dt=0.004, temperature=300.0, seed=seed, **kw)


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max()


@pytest.mark.parametrize("precision,tol", [(1, 1e-10), (0, 1e-5)])
def test_energy_and_forces_with_reciprocal_space(Engine, oracle_mod, tol_box, precision, tol):
    s, v = tol_box
    r = systems.with_reciprocal_space(s)
    data = _integ().to_data(precision=precision)
    g, o = Engine(r, data), oracle_mod.Oracle(r, data)
    for (ls, le) in ((1.0, 1.0), (0.5, 0.3), (0.0, 0.0)):
        eo, fo, to = o.energy_forces(ls, le)
        g.set_global("lambda_sterics", ls); g.set_global("lambda_electrostatics", le)
        tg = g.energy_terms()
        assert abs(to[8]) > 100.0 and abs(to[9]) > 10.0
        for k in range(10):
            assert abs(tg[k] - to[k]) <= tol * max(abs(to[k]), 1.0), (k, tg[k], to[k])
        assert abs(tg.sum() - eo) <= tol * abs(eo)
        assert g.potential_energy() == pytest.approx(tg.sum(), rel=1e-14)
        assert _rel(g.get_forces(), fo) <= tol
    g.close()


@pytest.mark.parametrize("precision,tol", [(1, 1e-9), (0, 1e-5)])
def test_s23k_frozen_environment_static_and_mobile_meshes(Engine, oracle_mod, precision, tol):
    """The frozen charges' meshes are computed once, the mobile ones per evaluation: their sum must be the full mesh result --
    checked on the forces of the mobile atoms and the energy, before and after the mobile atoms moved."""
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    r = systems.with_reciprocal_space(s)
    assert r.pme_grid == (18, 27, 36)
    data = _integ().to_data(precision=precision)
    g, o = Engine(r, data), oracle_mod.Oracle(r, data)
    mob = s.mass > 0
    for rnd in range(2):
        eo, fo, to = o.energy_forces(1.0, 1.0)
        tg = g.energy_terms()
        assert abs(tg[8] - to[8]) <= tol * abs(to[8]) and abs(tg[9] - to[9]) <= 1e-12 * abs(to[9])
        assert abs(tg.sum() - eo) <= tol * abs(eo)
        assert _rel(g.get_forces()[mob], fo[mob]) <= tol
        x = o.get_positions(); rng = np.random.RandomState(3)
        x[mob] += rng.normal(scale=0.004, size=(int(mob.sum()), 3))          # (constraints are not enforced by an evaluation)
        g.set_positions(x); o.set_positions(x)
    g.close()


@pytest.mark.parametrize("precision,tol", [(1, 1e-9), (0, 1e-5)])
def test_switch_with_reciprocal_space(Engine, oracle_mod, tol_box, precision, tol):
    s, v = tol_box
    r = systems.with_reciprocal_space(s)
    n = 20
    data = _integ(n).to_data(precision=precision)
    g, o = Engine(r, data), oracle_mod.Oracle(r, data)
    g.set_velocities(v); o.set_velocities(v)
    wg = g.run_switch(n, trace=True)
    wo = []
    for _ in range(n):
        o.step(1); wo.append(o.get_global("protocol_work"))
    wo = np.array(wo)
    assert np.abs(wg - wo).max() <= tol * np.abs(wo).max()
    assert np.abs(g.get_positions() - o.get_positions()).max() < (1e-9 if precision else 2e-4)
    # the trajectory is not the direct-space one (the forces on the water differ), the work of the first lambda step is (same x0)
    od = oracle_mod.Oracle(s, data); od.set_velocities(v); od.step(n)
    assert abs(od.get_global("protocol_work") - wo[-1]) > 1e-4
    g.close()


def test_batch_with_reciprocal_space_is_bitwise_solo(Engine, tune):
    from blues_amd.engine import NativeBatch
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    r = systems.with_reciprocal_space(s)
    R, n = 4, 12
    tune(assume_batch=8)   # separate force kernels, per-atom lists: the decomposition of a large batch, for the lone engines too

    def make():
        out = []
        for q in range(R):
            e = Engine(r, _integ(n, seed=40 + q).to_data(precision=0, replica=q)); e.set_velocities(v * (1 + 0.02 * q)); out.append(e)
        return out
    solo = make(); ws = [e.run_switch(n, trace=True) for e in solo]
    bat = make(); B = NativeBatch(bat); _, wb = B.step(n, trace=True)
    for q in range(R):
        assert np.array_equal(wb[q], ws[q]) and np.array_equal(solo[q].get_positions(), bat[q].get_positions())
        assert bat[q].potential_energy() == solo[q].potential_energy()
    B.close()


def test_alchemical_correction_is_no_longer_zero(Engine, oracle_mod, tol_box):
    """_computeAlchemicalCorrection (reference blues/simulation.py:1100-1119) compares the alchemical system at lambda = 1 with the
    MD system: under 'direct-space' PME treatment they differ by the ligand's share of reciprocal space (and of the dispersion
    correction).  Engine and oracle must agree on that difference."""
    s, v = tol_box
    r = systems.with_reciprocal_space(s)
    md = copy.copy(r); md.alchemical_atoms = np.zeros(0, np.int32)
    data = _integ().to_data(precision=1)
    ga, gm = Engine(r, data), Engine(md, data)
    oa, om = oracle_mod.Oracle(r, data), oracle_mod.Oracle(md, data)
    dg = ga.potential_energy() - gm.potential_energy()
    do = oa.energy_forces(1.0, 1.0)[0] - om.energy_forces(1.0, 1.0)[0]
    assert abs(do) > 0.05 and dg == pytest.approx(do, rel=1e-7, abs=1e-7)
    # without reciprocal space the two systems coincide at lambda = 1
    assert Engine(s, data).potential_energy() == pytest.approx(oracle_mod.Oracle(s, data).energy_forces(1.0, 1.0)[0], rel=1e-10)
    ga.close(); gm.close()


@pytest.mark.parametrize("reciprocal", [False, True])
def test_differential_correction_equals_the_four_energy_form(Engine, tol_box, reciprocal, tune):
    """SURVEY.md 8f.3 for the arrangement the reference actually runs -- md, alch and ncmc Simulations per chain
    (/root/reference/blues/simulation.py:768-809; the alch context wraps the MD System, :791-792): the alchemical correction
    -[(U_ncmc - U_md)(x0) + (U_alch - U_ncmc)(x1)] / kT (:1100-1119) formed as (D(x0) - D(x1)) / kT, D = U_alch - U_ncmc(lambda = 1) from the
    terms in which the two Systems differ -- nothing without reciprocal space; with PME the ligand's share of the mesh energy (the NCMC
    engine's own mesh kernel run with and without the alchemical charges: blues_mesh_energy), the erf corrections of the ligand's excluded
    pairs and constants of the box -- against the reference's four total energies, two of them in another context.  Double precision,
    whole BLUES iterations with MD legs, batched and chain by chain: the same corrections to 1e-9 of the energies they are differences of,
    the same decisions, the same states."""
    import copy as _copy
    from blues_amd import moves, simulation, unit
    from blues_amd.context import Simulation
    s, v = tol_box
    if reciprocal:
        s = systems.with_reciprocal_space(s)
    md_sys = _copy.copy(s); md_sys.alchemical_atoms = np.zeros(0, np.int32)
    plan = systems.alchemical_difference_plan(s, md_sys)
    assert plan is not None and plan["kind"] == ("pme" if reciprocal else "zero")
    lig = np.arange(15)
    R, nsteps, nmd, nIter = 3, 10, 6, 2
    tune(assume_batch=R)

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

    results = {}
    for differential in (False, None):
        for batched in (True, False):
            cs = chains(differential)
            assert all((c._diff_plan is not None) == (differential is None) for c in cs)
            B = simulation.BatchedBLUESSimulation(cs, batched_boundary=batched)
            assert B._differential() == (differential is None)
            records = []
            B.run(nIter=nIter, on_iteration=lambda N, last: records.append([dict(l) for l in last]))
            e_scale = max(abs(c.stateTable["md"]["state0"]["potential_energy"]._value) for c in cs)
            results[(differential, batched)] = (records, [c._md_sim.context._engine.get_positions() for c in cs], e_scale)
            B.close()
    kT = 0.0083144626 * 300.0
    ref = results[(False, True)]
    if reciprocal:   # (the correction is not zero: the ligand's share of reciprocal space changes between x0 and x1)
        assert max(abs(ref[0][N][r]["correction"]) for N in range(nIter) for r in range(R)) > 1e-4
    for key in ((None, True), (None, False), (False, False)):
        rec, xs, _ = results[key]
        for N in range(nIter):
            for r in range(R):
                a, b = rec[N][r], ref[0][N][r]
                # (1e-9 of the energies the correction is a difference of: a 10-step switch in this small box can end in a clash -- "the step is
                # unstable", protocol work 10^5 - 10^6 kJ/mol -- and the energies at x1 are then of that size, not of the start state's)
                assert abs(a["correction"] - b["correction"]) <= 1e-9 * max(ref[2], abs(b["protocol_work"])) / kT, (key, N, r, a["correction"], b["correction"], b["protocol_work"])
                assert a["accept"] == b["accept"] and a["protocol_work"] == pytest.approx(b["protocol_work"], rel=1e-9, abs=1e-9)
        for r in range(R):
            assert np.abs(xs[r] - ref[1][r]).max() < 1e-9
    if not reciprocal:   # the direct-space model: identically zero, no energy of another context evaluated
        assert all(results[(None, True)][0][N][r]["correction"] == 0.0 for N in range(nIter) for r in range(R))
