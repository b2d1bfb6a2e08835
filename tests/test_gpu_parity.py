"""GPU suite: the HIP engine, called through the C-ABI (blues_amd.engine.NativeEngine is a ctypes
shim over include/blues_engine.h), against the CPU oracle on identical inputs.

Tolerances: north_star asks for forces / energies / protocol work within 1e-5 relative of the
fp64 CPU reference.  precision="double" must agree to ~1e-10 (same algorithm, fp64 everywhere);
precision="mixed" (f32 pair math, f64 accumulation) must meet 1e-5 for single evaluations.  Long
trajectories are chaotic, so mixed-precision work traces are compared over short switches and the
full-length protocol is checked in double.
"""
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


def _integ(nsteps=20, dt=0.004, seed=7, **kw):
    return integrators.generateNCMCIntegrator(nstepsNC=nsteps, dt=dt, temperature=300.0, seed=seed, **kw)


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-300, np.abs(np.asarray(b)).max())


@pytest.mark.parametrize("precision,etol,ftol", [(1, 1e-10, 1e-10), (0, 1e-5, 1e-5)])
def test_energy_and_forces_parity_tol_box(Engine, oracle_mod, tol_box, precision, etol, ftol):
    s, v = tol_box
    data = _integ().to_data(precision=precision)
    g, o = Engine(s, data), oracle_mod.Oracle(s, data)
    for (ls, le) in ((1.0, 1.0), (0.5, 0.3), (0.05, 0.0), (0.0, 0.0)):
        eo, fo, to = o.energy_forces(ls, le)
        g.set_global("lambda_sterics", ls); g.set_global("lambda_electrostatics", le)
        tg = g.energy_terms()
        assert abs(tg.sum() - eo) <= etol * abs(eo)
        for k in range(8):
            assert abs(tg[k] - to[k]) <= etol * max(abs(to[k]), 1.0), (k, tg[k], to[k])
        assert g.potential_energy() == pytest.approx(tg.sum(), rel=1e-14)
        fg = g.get_forces()
        assert _rel(fg, fo) <= ftol
        assert np.linalg.norm(fg - fo) / np.linalg.norm(fo) <= ftol
    g.close()


@pytest.mark.parametrize("precision,tol", [(1, 1e-9), (0, 1e-5)])
def test_s23k_parity_frozen_and_full(Engine, oracle_mod, precision, tol):
    for frozen in (True, False):
        s, v = systems.s23k(frozen=frozen, restrained=0 if frozen else 40)
        data = _integ().to_data(precision=precision)
        g, o = Engine(s, data), oracle_mod.Oracle(s, data)
        for (ls, le) in ((1.0, 1.0), (0.3, 0.0)):
            eo, fo, to = o.energy_forces(ls, le)
            g.set_global("lambda_sterics", ls); g.set_global("lambda_electrostatics", le)
            tg = g.energy_terms()
            assert abs(tg.sum() - eo) <= tol * abs(eo)
            for k in range(8):
                assert abs(tg[k] - to[k]) <= tol * max(abs(to[k]), 1.0), (k, tg[k], to[k])
            fg = g.get_forces()
            mob = s.mass > 0
            assert _rel(fg[mob], fo[mob]) <= tol
            assert np.all(fg[~mob] == 0.0)  # forces on frozen atoms are never used by the path and are not computed
        g.close()


def _trace(o, n):
    w = []
    for _ in range(n):
        o.step(1); w.append(o.get_global("protocol_work"))
    return np.array(w)


@pytest.mark.parametrize("precision,tol", [(1, 1e-9), (0, 1e-5)])
def test_short_switch_work_trace(Engine, oracle_mod, tol_box, precision, tol):
    s, v = tol_box
    n = 20
    data = _integ(n).to_data(precision=precision)
    g, o = Engine(s, data), oracle_mod.Oracle(s, data)
    g.set_velocities(v); o.set_velocities(v)
    assert g.get_global("protocol_work") == 0.0
    wg = g.run_switch(n, trace=True)
    wo = _trace(o, n)
    scale = np.abs(wo).max()
    assert np.abs(wg - wo).max() <= tol * scale
    assert g.get_global("lambda") == pytest.approx(1.0) and g.get_global("step") == n
    assert g.get_global("protocol_work") == pytest.approx(wg[-1], rel=1e-15)
    ptol = 1e-9 if precision else 2e-4
    assert np.abs(g.get_positions() - o.get_positions()).max() < ptol
    # step() beyond nsteps is a no-op (reference blues/integrators.py:183)
    g.step(2)
    assert g.get_global("protocol_work") == pytest.approx(wg[-1], rel=1e-15) and g.get_global("step") == n
    # step(1) x n == run_switch(n) up to rounding: the fused launch takes the centre-of-mass momentum from the per-block
    # partials k_finalize leaves behind (sum m v + hV sum F), the call-by-call path from the kicked velocities themselves
    g2 = Engine(s, data); g2.set_velocities(v)
    for _ in range(n):
        g2.step(1)
    assert g2.get_global("protocol_work") == pytest.approx(wg[-1], rel=1e-11)
    assert np.abs(g2.get_positions() - g.get_positions()).max() < (1e-11 if precision else 1e-9)
    g.close(); g2.close()


@pytest.mark.parametrize("precision,seg,tol", [(1, 100, 1e-7), (0, 10, 1e-5)])
def test_full_length_protocol_teacher_forced(Engine, oracle_mod, tol_box, precision, seg, tol):
    """1000-step switch (BASELINE.json configs[1] length).  A liquid is chaotic (errors grow ~e^(7/ps t), measured),
    so two correct fp64 implementations drift apart pointwise over 4 ps; the whole lambda protocol is therefore
    checked segment by segment: the GPU state is re-synchronised to the oracle's at each segment start and the
    protocol-work increments of every segment must agree to `tol` of the work scale."""
    s, v = tol_box
    n = 1000
    data = _integ(n, seed=21).to_data(precision=precision)
    g, o = Engine(s, data), oracle_mod.Oracle(s, data)
    g.set_velocities(v); o.set_velocities(v)
    scale, worst = 0.0, 0.0
    wg_prev = wo_prev = 0.0
    for start in range(0, n, seg):
        if start:
            g.set_positions(o.get_positions()); g.set_velocities(o.get_velocities())
        wg = g.run_switch(seg, trace=True)
        wo = _trace(o, seg)
        dg = np.diff(np.concatenate([[wg_prev], wg])); do = np.diff(np.concatenate([[wo_prev], wo]))
        if start:  # the re-sync is itself an "instantaneous move": its work U(x_oracle)-U(x_gpu) lands in the first increment
            dg[0] -= g.get_global("perturbed_pe") - g.get_global("unperturbed_pe")
        worst = max(worst, np.abs(np.cumsum(dg) - np.cumsum(do)).max())
        scale = max(scale, np.abs(wo).max())
        wg_prev, wo_prev = wg[-1], wo[-1]
    assert scale > 10.0
    assert worst <= tol * scale, (worst, scale)
    assert g.get_global("lambda") == pytest.approx(1.0) and g.get_global("step") == n
    g.close()


def test_frozen_s23k_switch_with_move(Engine, oracle_mod):
    """configs[0]-style: 100-step switch of the S23k box (276 mobile atoms) with a rigid ligand move at lambda=0.5.
    The work of the instantaneous move is accounted (reference blues/integrators.py:184-191)."""
    s, v = systems.s23k(mobile_atoms=275)
    n = 20  # the oracle costs ~1 s per step at this size
    data = _integ(n, seed=5).to_data(precision=1)
    g, o = Engine(s, data), oracle_mod.Oracle(s, data)
    g.set_velocities(v); o.set_velocities(v)
    wg1 = g.run_switch(n // 2, trace=True); wo1 = _trace(o, n // 2)
    xg, xo = g.get_positions(), o.get_positions()
    assert np.abs(xg - xo).max() < 1e-9
    lig = np.asarray(s.alchemical_atoms)
    c = xo[lig].mean(0)
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    xn = xo.copy(); xn[lig] = (xo[lig] - c) @ R.T + c + np.array([0.02, 0.0, -0.01])
    g.set_positions(xn); o.set_positions(xn)
    wg2 = g.run_switch(n // 2, trace=True); wo2 = _trace(o, n // 2)
    wo, wg = np.concatenate([wo1, wo2]), np.concatenate([wg1, wg2])
    assert np.abs(wg - wo).max() <= 1e-8 * np.abs(wo).max()
    st = g.stats()
    assert st["i_tiles"] <= 6 and st["clusters"] <= 4 * 64  # only mobile atoms are integrated (cluster slots are padded per kind to wave boundaries)
    g.close()


@pytest.mark.parametrize("kw", [dict(nprop=2, propLambda=0.3), dict(splitting="R V O H O V R"), dict(splitting="V H R O R H V"),
                                 dict(splitting="O V R H R V O")])
def test_other_programs(Engine, oracle_mod, tol_box, kw):
    """Extra propagation inside the lambda window (reference blues/integrators.py:194-201) and other splittings."""
    s, v = tol_box
    n = 10
    kw = dict(kw)
    if "splitting" in kw:
        integ = integrators.AlchemicalExternalLangevinIntegrator(integrators.DEFAULT_ALCHEMICAL_FUNCTIONS, splitting=kw["splitting"],
                                                                  temperature=300.0, timestep=0.002, nsteps_neq=n, seed=13)
    else:
        integ = integrators.generateNCMCIntegrator(nstepsNC=n, dt=0.002, temperature=300.0, seed=13, **kw)
    data = integ.to_data(precision=1)
    g, o = Engine(s, data), oracle_mod.Oracle(s, data)
    g.set_velocities(v); o.set_velocities(v)
    wg = g.run_switch(n, trace=True); wo = _trace(o, n)
    assert np.abs(wg - wo).max() <= 1e-9 * max(1.0, np.abs(wo).max())
    assert np.abs(g.get_positions() - o.get_positions()).max() < 1e-10
    assert np.abs(g.get_velocities() - o.get_velocities()).max() < 1e-8
    g.close()


def test_cm_motion_removal_and_restraint(Engine, oracle_mod, tol_box):
    s, v = tol_box
    s2 = copy.copy(s); s2.remove_cm_motion = False
    s3 = systems.restrain_positions(s, np.arange(15, 975, 90), 2092.0)
    for sys_ in (s2, s3):
        data = _integ(8, seed=3).to_data(precision=1)
        g, o = Engine(sys_, data), oracle_mod.Oracle(sys_, data)
        g.set_velocities(v); o.set_velocities(v)
        wg = g.run_switch(8, trace=True); wo = _trace(o, 8)
        assert np.abs(wg - wo).max() <= 1e-9 * np.abs(wo).max()
        assert np.abs(g.get_velocities() - o.get_velocities()).max() < 1e-8
        g.close()
    # with the remover on, the centre-of-mass momentum of the mobile atoms is ~0 at the start of every pass
    data = _integ(4, seed=3).to_data(precision=1)
    g = Engine(s, data); g.set_velocities(v + 0.3); g.step(1)
    g.close()


def test_rng_and_velocity_initialisation(Engine, oracle_mod, tol_box):
    s, _ = tol_box
    data = _integ().to_data(precision=1)
    g, o = Engine(s, data), oracle_mod.Oracle(s, data)
    g.set_velocities_to_temperature(300.0, 424242); o.set_velocities_to_temperature(300.0, 424242)
    vg, vo = g.get_velocities(), o.get_velocities()
    assert np.abs(vg - vo).max() < 1e-10
    ndof = 3 * s.n_atoms - len(s.constraint_dist)
    T = 2 * g.kinetic_energy() / (ndof * 0.0083144626)
    assert 270 < T < 330 and g.kinetic_energy() == pytest.approx(o.kinetic_energy(), rel=1e-12)
    # velocity constraints hold after initialisation (OpenMM applies them in setVelocitiesToTemperature)
    x, c = s.positions, s.constraint_atoms
    assert np.abs(((vg[c[:, 0]] - vg[c[:, 1]]) * (x[c[:, 0]] - x[c[:, 1]])).sum(1)).max() < 1e-10
    g.close()


def test_determinism_reset_and_globals(Engine, tol_box):
    s, v = tol_box
    data = _integ(12, seed=99).to_data(precision=0)
    runs = []
    for _ in range(2):
        g = Engine(s, data); g.set_velocities(v)
        w = g.run_switch(12, trace=True)
        runs.append((w, g.get_positions(), g.get_velocities()))
        # reset() zeroes the integrator globals but not the context (reference blues/integrators.py:240-249)
        g.reset()
        assert g.get_global("step") == 0 and g.get_global("protocol_work") == 0.0 and g.get_global("lambda") == 0.0
        assert g.get_global("first_step") == 0 and g.get_global("prop") == 1
        # the afterMove rejection device (reference blues/moves.py:1082, tests/test_watertranslation.py:112)
        g.set_global("protocol_work", 999999)
        assert g.get_global("protocol_work") >= 999999
        with pytest.raises(Exception, match="unknown global"):
            g.get_global("no_such_variable")
        g.close()
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])


def test_error_behaviour(Engine, tol_box):
    from blues_amd.engine import EngineError
    s, v = tol_box
    g = Engine(s, _integ(4).to_data())
    x = s.positions.copy(); x[100] = np.nan
    g.set_positions(x); g.set_velocities(v)
    with pytest.raises(EngineError, match="nan"):
        g.step(1)
    g.close()
    s2 = copy.copy(s); s2.mass = s.mass.copy(); s2.mass[15] = 0.0
    with pytest.raises(EngineError, match="massless"):
        Engine(s2, _integ(4).to_data())
    s3 = copy.copy(s); s3.box = np.array([1.5, 2.1786, 2.1786])
    with pytest.raises(EngineError, match="cutoff"):
        Engine(s3, _integ(4).to_data())


def test_full_size_properties(Engine):
    """Size-independent properties at BASELINE.json's full size (all 23,400 atoms mobile, configs[3]-like)."""
    s, v = systems.s23k(frozen=False)
    data = _integ(10, seed=8).to_data(precision=0)
    g = Engine(s, data)
    e0, f0 = g.potential_energy(), g.get_forces()
    assert np.abs(f0.sum(0)).max() < 1e-5 * np.abs(f0).max() * np.sqrt(s.n_atoms)  # Newton's third law (f32 pair rounding only)
    # rigid translation and a lattice translation leave energies and forces unchanged
    g.set_positions(s.positions + np.array([0.37, -1.2, 2.9]))
    assert g.potential_energy() == pytest.approx(e0, rel=2e-7)
    g.set_positions(s.positions + np.array([s.box[0], 0.0, -s.box[2]]))
    assert g.potential_energy() == pytest.approx(e0, rel=1e-9)
    assert _rel(g.get_forces(), f0) < 1e-6
    # a permutation of the atom order (here: whole-molecule reversal of the water block) changes nothing physical
    g.set_positions(s.positions); g.set_velocities(v)
    w = g.run_switch(10, trace=True)
    assert np.all(np.isfinite(w)) and g.stats()["i_tiles"] == (s.n_atoms - 15 + 63) // 64
    x = g.get_positions(); c = s.constraint_atoms
    assert np.abs(np.linalg.norm(x[c[:, 0]] - x[c[:, 1]], axis=1) / s.constraint_dist - 1).max() < 1e-8
    assert g.time_nonbonded(5) > 0
    g.close()


def test_context_surface_on_gpu(tol_box):
    """The OpenMM-like objects BLUES consumes, bound to the real engine."""
    from blues_amd import moves, simulation, unit
    from blues_amd.context import Simulation
    s, v = tol_box
    integ = _integ(10, seed=4)
    sim = Simulation(None, s, integ, platformProperties={"DeviceIndex": 0, "Precision": "mixed"})
    sim.context.setVelocities(unit.Quantity(v, "nanometer/picosecond"))
    lig = np.arange(15)
    b = simulation.BLUESSimulation(simulation.SimulationSet(sim), {"nstepsNC": 10, "moveStep": 5, "nIter": 2},
                                   moves.MoveEngine(moves.RandomLigandRotationMove(lig, s.mass[lig], random_state=1)))
    np.random.seed(1)
    b.run()
    assert b.accept + b.reject == 2
    assert np.isfinite(b.last["log_accept"]) and abs(b.last["correction"]) < 1e-6
    assert sim.context.getPlatform().getName().startswith("HIP")


@pytest.mark.parametrize("precision,tol", [(1, 1e-9), (0, 1e-5)])
def test_committed_golden_vectors(Engine, tol_box, precision, tol):
    """The engine against the committed vectors of tests/golden/tol_box_oracle_vectors.json (no oracle at run time)."""
    import json, os
    s, v = tol_box
    with open(os.path.join(os.path.dirname(__file__), "golden", "tol_box_oracle_vectors.json")) as fh:
        vec = json.load(fh)
    ic = vec["integrator"]
    data = integrators.generateNCMCIntegrator(nstepsNC=ic["nstepsNC"], dt=ic["dt"], temperature=ic["temperature"], seed=ic["seed"]).to_data(precision=precision)
    g = Engine(s, data); g.set_velocities(v)
    for rec in vec["energies"]:
        g.set_global("lambda_sterics", rec["lambda_sterics"]); g.set_global("lambda_electrostatics", rec["lambda_electrostatics"])
        t = g.energy_terms()
        assert abs(t.sum() - rec["total"]) <= tol * abs(rec["total"])
        f = g.get_forces()
        fr = np.array(rec["forces"])
        assert np.abs(f[rec["force_atoms"]] - fr).max() <= tol * rec["force_norm"] / np.sqrt(s.n_atoms) * 30
    g.set_global("lambda_sterics", 1.0); g.set_global("lambda_electrostatics", 1.0)
    w = g.run_switch(ic["nstepsNC"], trace=True)
    wref = np.array(vec["work_trace"])
    assert np.abs(w - wref).max() <= tol * np.abs(wref).max()
    g.close()


def test_md_leg_openmm_langevin(Engine, oracle_mod, tol_box):
    """The MD leg: OpenMM's LangevinIntegrator step on the non-alchemical system, all atoms mobile."""
    s, v = tol_box
    md = copy.copy(s); md.alchemical_atoms = np.zeros(0, np.int32)
    integ = integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=17)
    integ.setConstraintTolerance(1e-8)
    data = integ.to_data(precision=1)
    g, o = Engine(md, data), oracle_mod.Oracle(md, data)
    g.set_velocities(v); o.set_velocities(v)
    assert g.potential_energy() == pytest.approx(o.energy_forces(1.0, 1.0)[0], rel=1e-12)
    g.step(25); o.step(25)
    assert np.abs(g.get_positions() - o.get_positions()).max() < 1e-9
    assert np.abs(g.get_velocities() - o.get_velocities()).max() < 1e-7
    x, c = g.get_positions(), md.constraint_atoms
    assert np.abs(np.linalg.norm(x[c[:, 0]] - x[c[:, 1]], axis=1) / md.constraint_dist - 1).max() < 1e-8
    ndof = 3 * md.n_atoms - len(md.constraint_dist)
    T = 2 * g.kinetic_energy() / (ndof * 0.0083144626)
    assert 250 < T < 350
    g.close()


def test_full_blues_iteration_on_gpu(tol_box):
    """NCMC leg + Metropolis + MD leg, all on the engine, through the driver mirror (reference BLUESSimulation.run)."""
    from blues_amd import moves, simulation, unit
    from blues_amd.context import Simulation
    s, v = tol_box
    md_sys = copy.copy(s); md_sys.alchemical_atoms = np.zeros(0, np.int32)
    ncmc = Simulation(None, s, _integ(10, seed=4))
    md = Simulation(None, md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=5))
    alch = Simulation(None, md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=6))
    for sim in (ncmc, md, alch):
        sim.context.setVelocities(unit.Quantity(v, "nanometer/picosecond"))
    lig = np.arange(15)
    b = simulation.BLUESSimulation(simulation.SimulationSet(ncmc, md=md, alch=alch), {"nstepsNC": 10, "moveStep": 5, "nIter": 2, "nstepsMD": 10},
                                   moves.MoveEngine(moves.RandomLigandRotationMove(lig, s.mass[lig], random_state=2)))
    np.random.seed(3)
    x0 = md.context.getState(getPositions=True).getPositions(asNumpy=True)._value.copy()
    # a 10-step protocol is far too fast for a 15-atom ligand: the random rotation lands on waters, the work explodes
    # and the switch may even become unstable -- the engine then raises like OpenMM would, the driver logs it, calls
    # move._error and carries on (reference blues/simulation.py:1088-1094); either way the move must end up rejected
    b.run()
    assert b.accept + b.reject == 2 and md.currentStep == 20 and b.accept == 0
    # U_md == U_alch(lambda=1) in the direct-space-only model -- to the rounding of the mixed-precision pair sums: the two contexts
    # evaluate the ligand's pairs on different kernels (fp64 alchemical kernel / fp32 pair kernel), 1e-9 of the 12,000 kJ/mol total
    assert abs(b.last["correction"]) < 1e-4
    x1 = md.context.getState(getPositions=True).getPositions(asNumpy=True)._value
    assert np.not_equal(x0, x1).all()          # reference blues/tests/test_simulation.py: positions change after _stepMD


@pytest.mark.parametrize("precision,tol", [(1, 1e-9), (0, 1e-5)])
def test_alchemical_subset_of_a_molecule(Engine, oracle_mod, tol_box, precision, tol):
    """configs[4]-style region: only PART of the solute is alchemical (here the methyl group of toluene), so there are
    exclusions and 1-4 exceptions between alchemical and environment atoms (openmmtools moves those to scaled
    CustomBondForces -- SURVEY.md Appendix B); annihilate_sterics on/off both covered."""
    s, v = tol_box
    for annih in (False, True):
        sub = copy.copy(s); sub.alchemical_atoms = np.array([0, 7, 8, 9], np.int32); sub.annihilate_sterics = annih
        data = _integ(12, seed=31).to_data(precision=precision)
        g, o = Engine(sub, data), oracle_mod.Oracle(sub, data)
        for (ls, le) in ((1.0, 1.0), (0.4, 0.25), (0.0, 0.0)):
            eo, fo, to = o.energy_forces(ls, le)
            g.set_global("lambda_sterics", ls); g.set_global("lambda_electrostatics", le)
            tg = g.energy_terms()
            for k in range(8):
                assert abs(tg[k] - to[k]) <= tol * max(abs(to[k]), 1.0), (annih, ls, le, k, tg[k], to[k])
            assert _rel(g.get_forces(), fo) <= tol
        g.set_global("lambda_sterics", 1.0); g.set_global("lambda_electrostatics", 1.0)
        g.set_velocities(v); o.set_velocities(v)
        wg = g.run_switch(12, trace=True); wo = _trace(o, 12)
        assert np.abs(wg - wo).max() <= tol * max(1.0, np.abs(wo).max())
        g.close()


def test_thermostat_and_constraints_long_run(Engine, tol_box):
    """Physics check that does not involve the oracle: 3000 Langevin steps (12 ps) of the 975-atom box at 4 fs keep the
    kinetic temperature at 300 K (equipartition over 3N - Nc degrees of freedom), the constraints at 1e-8 and the
    centre-of-mass momentum of the CM-removed system at zero -- for both the BAOAB-like 'V R O R V' and OpenMM's 'L' step."""
    s, v = tol_box
    ndof = 3 * s.n_atoms - len(s.constraint_dist) - 3
    for split in ("V R O R V", "L"):
        integ = integrators.AlchemicalExternalLangevinIntegrator({"lambda_sterics": "1", "lambda_electrostatics": "1"}, splitting="V R O R V",
                                                                  temperature=300.0, timestep=0.004, nsteps_neq=2 ** 30, seed=77) if split != "L" else \
            integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=78)
        g = Engine(s, integ.to_data(precision=0)); g.set_velocities(v)
        temps = []
        for blk in range(60):
            g.step(100)
            temps.append(2 * g.kinetic_energy() / (ndof * 0.0083144626))
        T = np.mean(temps[5:])
        # 975 atoms: one sample scatters by 300 K * sqrt(2 / ndof) = 9 K and the samples are correlated over ~0.5 ps, so the mean of
        # this window carries ~2 K of noise on top of the discretisation offset at 4 fs (both schemes read 3-4 K low over
        # 60 ps runs; OpenMM's leapfrog-style 'L' does so in OpenMM itself)
        assert abs(T - 300.0) < 9.0, (split, T)
        x, vv, c = g.get_positions(), g.get_velocities(), s.constraint_atoms
        assert np.abs(np.linalg.norm(x[c[:, 0]] - x[c[:, 1]], axis=1) / s.constraint_dist - 1).max() < 1e-7
        # the remover zeroes the total momentum at the head of every pass; what is left at the end of a step is the
        # momentum injected by that step's thermostat noise: sqrt(sum m kT (1 - exp(-2 gamma dt))) ~ 12 amu nm/ps here
        p = (s.mass[:, None] * vv).sum(0)
        assert np.abs(p).max() < 6.0 * np.sqrt(s.mass.sum() * 0.0083144626 * 300.0 * (1.0 - np.exp(-2 * 0.004)))
        assert g.stats()["list_generation"] > 5        # the lists were rebuilt on the device along the way
        g.close()


def _driver_run(engine_factory, system, vel, move_factory, nsteps, seed, precision="double", dt=0.002):
    """One _stepNCMC + Metropolis test through the driver mirror on the given engine; returns the decision record."""
    from unittest import mock
    from blues_amd import context, moves, simulation, unit
    integ = _integ(nsteps, dt=dt, seed=seed)
    # engine_factory=None -> the shipped HIP engine; otherwise the test double replaces it for the duration of the constructor only
    with mock.patch.object(context, "NativeEngine", engine_factory or context.NativeEngine):
        sim = context.Simulation(None, system, integ, precision=precision)
    sim.context.setVelocities(unit.Quantity(vel, "nanometer/picosecond"))
    np.random.seed(seed)
    b = simulation.BLUESSimulation(simulation.SimulationSet(sim), {"nstepsNC": nsteps, "moveStep": nsteps // 2, "nIter": 1}, moves.MoveEngine(move_factory()))
    b._syncStatesMDtoNCMC()
    b._stepNCMC(nsteps, nsteps // 2)
    b._acceptRejectMove()
    return b.last, b.stateTable["ncmc"]["state1"]["positions"]._value.copy(), b.stateTable["ncmc"]["state1"]["potential_energy"]._value


def test_driver_end_to_end_water_move(Engine, tol_box):
    """configs[3] in miniature: WaterTranslationMove (swap at step 0, sphere translation at lambda = 0.5, radius check at the
    end), nothing frozen, positional restraints -- the same driver with the same seeds on the HIP engine and on the oracle."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from conftest import OracleBackedEngine
    from test_moves_configs import water_system
    from blues_amd import moves
    w, v, waters = water_system(tol_box)
    mk = lambda: moves.WaterTranslationMove(waters, np.arange(15), w.mass[:15], radius=0.9)

    class Outside(moves.WaterTranslationMove):
        """Places the water 10 % beyond the sphere so that afterMove must force the rejection (reference blues/moves.py:1082);
        in the reference this happens when the water diffuses out during the second half of the protocol."""
        def _random_sphere_point(self, radius, origin):
            d = super()._random_sphere_point(radius, origin) - origin
            return origin + d / np.linalg.norm(d) * radius * 1.1
    mk_out = lambda: Outside(waters, np.arange(15), w.mass[:15], radius=0.9)
    # Short protocols that insert onto overlapping waters blow up physically (in OpenMM too); the seeds are ones where they do not.
    for factory, nsteps, seed in ((mk, 60, 15), (mk_out, 60, 9), (mk_out, 60, 7)):
        rg, xg, eg = _driver_run(None, w, v, factory, nsteps, seed=seed, dt=0.001)
        ro, xo, eo = _driver_run(OracleBackedEngine, w, v, factory, nsteps, seed=seed, dt=0.001)
        assert rg["accept"] == ro["accept"]
        assert rg["protocol_work"] == pytest.approx(ro["protocol_work"], rel=1e-8, abs=1e-7)
        assert rg["log_accept"] == pytest.approx(ro["log_accept"], rel=1e-8, abs=1e-7)
        assert np.abs(xg - xo).max() < 1e-8 and eg == pytest.approx(eo, rel=1e-10)
        assert (ro["protocol_work"] >= 999999) == (factory is mk_out)


def test_driver_end_to_end_rotation_and_torsion(Engine, tol_box):
    """configs[0]/[4] in miniature: ligand rotation on the fully alchemical toluene, and a methyl torsion move with only the
    methyl group alchemical (alchemical-environment exceptions active)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from conftest import OracleBackedEngine
    from blues_amd import moves
    s, v = tol_box
    lig = np.arange(15)
    mk = lambda: moves.RandomLigandRotationMove(lig, s.mass[lig], random_state=8)
    rg, xg, eg = _driver_run(None, s, v, mk, 12, seed=5)
    ro, xo, eo = _driver_run(OracleBackedEngine, s, v, mk, 12, seed=5)
    assert rg["accept"] == ro["accept"] and rg["protocol_work"] == pytest.approx(ro["protocol_work"], rel=1e-8, abs=1e-7)
    assert np.abs(xg - xo).max() < 1e-8
    sub = copy.copy(s); sub.alchemical_atoms = np.array([0, 7, 8, 9], np.int32)
    mk2 = lambda: moves.TorsionRotationMove((1, 0), [7, 8, 9], random_state=9)
    rg, xg, eg = _driver_run(None, sub, v, mk2, 12, seed=6)
    ro, xo, eo = _driver_run(OracleBackedEngine, sub, v, mk2, 12, seed=6)
    assert rg["accept"] == ro["accept"] and rg["protocol_work"] == pytest.approx(ro["protocol_work"], rel=1e-8, abs=1e-7)
    assert np.abs(xg - xo).max() < 1e-8 and eg == pytest.approx(eo, rel=1e-10)


def test_resort_when_tiles_spread(Engine, tune):
    """Nothing frozen (configs[3]): the tiles formed at set_positions spread as the liquid diffuses and their j-lists grow.
    With the list capacity squeezed, the engine must notice (resort_hint), re-sort on its own and carry on -- with the
    same trajectory as a run whose lists never came near their capacity (fp64 mode: equal to summation-order rounding)."""
    s, v = systems.s23k(frozen=False)
    n = 192
    data = _integ(n, seed=17).to_data(precision=1)
    g0 = Engine(s, data); g0.set_velocities(v)
    w0 = g0.run_switch(n, trace=True)
    assert g0.stats()["resorts"] == 0
    cap0 = g0.stats()["jcap"]; top = g0.stats()["max_jcount"]
    tune(jcap_scale=1.12 * top / cap0)   # capacity ~12 % above the longest list
    g1 = Engine(s, data); g1.set_velocities(v)
    assert g1.stats()["jcap"] < cap0
    w1 = g1.run_switch(n, trace=True)
    assert g1.stats()["resorts"] >= 1
    assert np.allclose(w1, w0, rtol=1e-9, atol=1e-8)
    assert np.abs(g1.get_positions() - g0.get_positions()).max() < 1e-9
    g0.close(); g1.close()


def test_barostat_on_gpu_matches_the_oracle(Engine, oracle_mod, tol_box):
    """MD leg with a MonteCarloBarostat (reference blues/simulation.py:603-626): the same volume moves (same barostat seed) on the HIP
    engine and on the oracle -- the two energies of every attempt come from the device after blues_set_box re-tiled the system for the new
    box, with reciprocal space on (mesh terms and the dispersion correction depend on the volume)."""
    from blues_amd import barostat
    import copy as _copy
    s, v = tol_box
    md = systems.with_reciprocal_space(s); md = _copy.copy(md); md.alchemical_atoms = np.zeros(0, np.int32)
    data = integrators.LangevinIntegrator(300.0, 1.0, 0.002, seed=9).to_data(precision=1)
    g, o = Engine(md, data), oracle_mod.Oracle(md, data)

    class OracleSide:   # the barostat needs get_box / set_box / get_positions / set_positions / potential_energy
        def __init__(self, orc, box): self.o, self.box = orc, np.diag(np.asarray(box, float))
        def get_box(self): return self.box.copy()
        def set_box(self, b): self.box = np.diag(np.asarray(b, float).reshape(-1)[:3]) if np.size(b) == 3 else np.asarray(b, float).reshape(3, 3); self.o.set_box(np.diag(self.box))
        def get_positions(self): return self.o.get_positions()
        def set_positions(self, x): self.o.set_positions(x)
        def potential_energy(self): return self.o.energy_forces(1.0, 1.0)[0]
    os_ = OracleSide(o, md.box)
    bg, bo = barostat.MonteCarloBarostat(1.0, 300.0, 25, seed=5), barostat.MonteCarloBarostat(1.0, 300.0, 25, seed=5)
    g.set_velocities(v); o.set_velocities(v)
    n_acc = 0
    for k in range(6):
        ag, ao = bg.attempt(g, md), bo.attempt(os_, md)
        assert ag == ao
        assert bg.last["dU"] == pytest.approx(bo.last["dU"], rel=1e-7, abs=1e-6) and bg.last["volume"] == pytest.approx(bo.last["volume"], rel=1e-14)
        assert np.allclose(np.diag(g.get_box()), np.diag(os_.get_box()), rtol=1e-14)
        assert g.potential_energy() == pytest.approx(os_.potential_energy(), rel=1e-10)
        n_acc += ag
        g.step(5); o.step(5)
        assert np.abs(g.get_positions() - o.get_positions()).max() < 1e-8
    assert 0 < n_acc
    g.close()


@pytest.mark.parametrize("precision", [1, 0])
@pytest.mark.parametrize("reciprocal", [False, True])
def test_unmodified_potential_comes_with_every_energy_evaluation(Engine, oracle_mod, tol_box, precision, reciprocal):
    """the energies _computeAlchemicalCorrection takes at lambda_sterics = lambda_electrostatics = 1 differ from the
    energy at the current parameters only in the alchemical terms.  Every energy evaluation carries those terms at (1, 1) in a spare
    slot of the same pass, so asking for the energy at (1, 1) afterwards costs no evaluation -- and returns, bit for bit, what a
    direct evaluation at (1, 1) returns."""
    s, v = tol_box
    if reciprocal:
        s = systems.with_reciprocal_space(s)
    data = _integ().to_data(precision=precision)
    g = Engine(s, data)
    g.set_global("lambda_sterics", 0.4); g.set_global("lambda_electrostatics", 0.25)
    e_mid = g.potential_energy()
    n0 = g.stats()["own_energy_evaluations"]
    g.set_global("lambda_sterics", 1.0); g.set_global("lambda_electrostatics", 1.0)
    e_one = g.potential_energy()
    assert g.stats()["own_energy_evaluations"] == n0, "the energy at (1, 1) should have come with the previous evaluation"
    g2 = Engine(s, data)
    assert g2.potential_energy() == e_one     # fresh engine: parameters default to (1, 1), a direct evaluation
    o = oracle_mod.Oracle(s, data)
    assert e_one == pytest.approx(o.energy_forces(1.0, 1.0)[0], rel=1e-10 if precision else 1e-6)
    assert e_mid == pytest.approx(o.energy_forces(0.4, 0.25)[0], rel=1e-10 if precision else 1e-6) and abs(e_mid - e_one) > 1.0
    g.step(1)                                  # positions moved: both cached values are gone
    g.set_global("lambda_sterics", 1.0); g.set_global("lambda_electrostatics", 1.0)
    assert g.potential_energy() != e_one
