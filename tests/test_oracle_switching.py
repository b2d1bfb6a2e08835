"""CPU suite: the oracle's restatement of the switching integrators of reference blues/switching.py (NCMCVVAlchemicalIntegrator
:1083-1241, NCMCGHMCAlchemicalIntegrator :1244-1360; dead code in the reference, so no reference-held vector exists) against facts
that hold for any correct implementation, and the host-side mirror classes (blues_amd/switching.py)."""
import numpy as np
import pytest

from blues_amd import switching, systems

FUNCS = {'lambda_sterics': 'lambda', 'lambda_electrostatics': 'lambda^0.5'}
KB = 0.0083144626


@pytest.fixture(scope="module")
def box():
    import copy
    s, v = systems.toluene_box()
    s = copy.copy(s)
    s.remove_cm_motion = False   # (the CMMotionRemover acts between the energy brackets: it would break the exact work balance below)
    return s, v


def _oracle(oracle_mod, box, integ):
    s, v = box
    o = oracle_mod.Oracle(s, integ.to_data())
    o.set_velocities(v)
    return o


@pytest.mark.parametrize("direction,psteps", [("insert", 1), ("delete", 2)])
def test_vv_total_work_is_the_change_of_total_energy(oracle_mod, box, direction, psteps):
    """Velocity Verlet has no thermostat: protocol work (parameter changes at fixed positions) + shadow work (energy drift of the
    propagation) must add up to the change of energy + kinetic between the initial alchemical state and the end."""
    it = switching.NCMCVVAlchemicalIntegrator(300.0, None, FUNCS, nsteps=4, steps_per_propagation=psteps, timestep=0.002, direction=direction)
    o = _oracle(oracle_mod, box, it)
    ke0 = o.kinetic_energy()   # fixture velocities satisfy the constraints, so the first-step projection leaves them alone
    o.step(4)
    g = o.get_global
    assert g("step") == 4.0 and g("lambda_step") == 4.0
    total = g("protocol_work") + g("shadow_work")
    assert g("total_work") == pytest.approx(total, rel=1e-14)
    assert abs(g("protocol_work")) > 10.0 and abs(g("shadow_work")) > 1e-3
    e_end = g("final_energy") + o.kinetic_energy()
    e_start = g("initial_energy") + ke0
    assert total == pytest.approx(e_end - e_start, abs=2e-6)
    # the end state is the other end of the path
    assert g("lambda_sterics") == (1.0 if direction == "insert" else 0.0)
    # further calls do nothing once step == nsteps (switching.py:1224)
    x = o.get_positions().copy(); o.step(2)
    assert np.array_equal(x, o.get_positions()) and g("step") == 4.0


def test_instantaneous_toggle(oracle_mod, box):
    """nsteps = 0 (switching.py:1198-1207): one perturbation from the initial to the final state, no propagation."""
    it = switching.NCMCVVAlchemicalIntegrator(300.0, None, FUNCS, nsteps=0, direction="delete")
    o = _oracle(oracle_mod, box, it)
    x = o.get_positions().copy()
    o.step(1)
    g = o.get_global
    e1 = o.energy_forces(1.0, 1.0)[0]; e0 = o.energy_forces(0.0, 0.0)[0]
    assert g("initial_energy") == pytest.approx(e1, rel=1e-12) and g("final_energy") == pytest.approx(e0, rel=1e-12)
    assert g("total_work") == pytest.approx(e0 - e1, rel=1e-10) and g("shadow_work") == 0.0
    assert np.abs(x - o.get_positions()).max() < 1e-6


def test_ghmc_counts_trials_and_is_reproducible(oracle_mod, box):
    def run(seed):
        it = switching.NCMCGHMCAlchemicalIntegrator(300.0, None, FUNCS, nsteps=5, timestep=0.002, direction="insert")
        it.setRandomNumberSeed(seed)
        o = _oracle(oracle_mod, box, it)
        o.step(5)
        return o
    a, b, c = run(11), run(11), run(12)
    g = a.get_global
    assert g("ntrials") == 6.0 and 0.0 <= g("naccept") <= 6.0        # one GHMC step in the first-step block + one per step
    assert g("shadow_work") == 0.0 and g("total_work") == g("protocol_work")   # GHMC books no shadow work (switching.py:961-1017)
    assert np.array_equal(a.get_positions(), b.get_positions()) and g("protocol_work") == b.get_global("protocol_work")
    assert not np.array_equal(a.get_positions(), c.get_positions())
    a.reset()
    assert g("ntrials") == 0.0 and g("total_work") == 0.0 and g("step") == 0.0


def test_ghmc_rejection_restores_positions_and_flips_momenta(oracle_mod, box):
    """With a timestep far too long every trial is rejected: positions must stay where the constraints put them, whatever the
    randomisations do to the velocities."""
    it = switching.NCMCGHMCAlchemicalIntegrator(300.0, None, {'lambda_sterics': '1', 'lambda_electrostatics': '1'}, nsteps=2, timestep=0.02,
                                                collision_rate=0.0, direction="insert")
    o = _oracle(oracle_mod, box, it)
    x0, v0 = o.get_positions().copy(), o.get_velocities().copy()
    o.step(1)
    g = o.get_global
    assert g("ntrials") == 2.0 and g("naccept") == 0.0
    assert np.abs(o.get_positions() - x0).max() < 1e-7
    # collision rate 0: the randomisations are the identity, two rejections flip the momenta twice
    assert np.abs(o.get_velocities() - v0).max() < 1e-6
    assert g("protocol_work") == 0.0   # nothing depends on lambda in this protocol


def test_mirror_classes_follow_the_reference_signatures(box):
    s, _ = box
    with pytest.raises(Exception, match="'direction' must be one of"):
        switching.NCMCVVAlchemicalIntegrator(300.0, s, FUNCS, nsteps=2, direction="sideways")
    vv = switching.NCMCVVAlchemicalIntegrator(300.0, s, FUNCS, nsteps=4, steps_per_propagation=3, timestep=0.001, direction="delete")
    assert vv.nsteps == 4 and vv.direction == "delete" and vv.has_statistics is False and vv.getStatistics(None) == (0, 0)
    assert vv.kT._value == pytest.approx(KB * 300.0)
    d = vv.to_data()
    assert d.switching_mode == 1 and d.steps_per_propagation == 3 and d.n_lambda_steps == 4 and d.timestep == 0.001
    assert np.allclose(d.lambda_sterics, [1.0, 0.75, 0.5, 0.25, 0.0]) and np.allclose(d.lambda_electrostatics, np.sqrt([1.0, 0.75, 0.5, 0.25, 0.0]))
    gh = switching.NCMCGHMCAlchemicalIntegrator(300.0, s, {'lambda_sterics': 'lambda', 'lambda_torsions': 'lambda'}, nsteps=2)
    assert gh.has_statistics and gh._collision_rate == 9.1 and gh._ignored_functions == ['lambda_torsions']
    dg = gh.to_data()
    assert dg.switching_mode == 2 and np.allclose(dg.lambda_sterics, [0.0, 0.5, 1.0]) and np.allclose(dg.lambda_electrostatics, 1.0)
    fl = switching.NCMCVVAlchemicalIntegrator(300.0, s, {'lambda_sterics': 'lambda'}, nsteps=2, direction="flux")
    assert np.allclose(fl.to_data().lambda_sterics, [1.0, 0.5, 1.0])   # reset to 1.0, then (step+1)/nsteps (switching.py:866, 899)
    assert vv.getLogAcceptanceProbability(None) == 0.0 and vv.get_step() == 0.0   # unbound: the reference's initial globals
