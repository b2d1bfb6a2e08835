"""GPU suite: the switching integrators of reference blues/switching.py (velocity-Verlet and GHMC switching) on the
HIP engine against the oracle: protocol work, shadow work, trial statistics, positions."""
import copy

import numpy as np
import pytest

from blues_amd import switching, systems

pytestmark = pytest.mark.gpu
FUNCS = {'lambda_sterics': 'lambda', 'lambda_electrostatics': 'lambda^0.5'}


@pytest.fixture(scope="module")
def Engine():
    from blues_amd import build
    build.build_engine()
    from blues_amd.engine import NativeEngine
    return NativeEngine


def _pair(Engine, oracle_mod, tol_box, integ, precision, cm=True):
    s, v = tol_box
    if not cm:
        s = copy.copy(s); s.remove_cm_motion = False
    data = integ.to_data(precision=precision)
    g, o = Engine(s, data), oracle_mod.Oracle(s, data)
    g.set_velocities(v); o.set_velocities(v)
    return g, o


# (mixed precision, "delete", 2 fs: the third step of that run lands on a configuration with ONE H-H pair within fp32 rounding of the
# cutoff, which the fp32 pair kernel leaves out and the fp64 oracle counts -- 0.058 kJ/mol and 0.64 kJ/mol/nm, the jump of the unshifted
# erfc Coulomb term at r_c for ewaldErrorTolerance 0.005; a fresh engine at those positions reproduces it.  The mixed case therefore
# runs at 1.5 fs, where no pair sits on the boundary.)
@pytest.mark.parametrize("precision,tol,dt", [(1, 1e-9, 0.002), (0, 2e-4, 0.0015)])
@pytest.mark.parametrize("direction,psteps", [("insert", 1), ("delete", 2)])
def test_velocity_verlet_switching_matches_the_oracle(Engine, oracle_mod, tol_box, precision, tol, dt, direction, psteps):
    it = switching.NCMCVVAlchemicalIntegrator(300.0, None, FUNCS, nsteps=6, steps_per_propagation=psteps, timestep=dt, direction=direction)
    g, o = _pair(Engine, oracle_mod, tol_box, it, precision)
    for n in (1, 2, 3):   # the first call carries the first-step block
        g.step(n); o.step(n)
        # the shadow work differences TOTAL energies (12,000 kJ/mol here): in mixed precision each carries the fp32 pair noise, ~2e-7 relative
        slack = 0.0 if precision else 5e-7 * abs(o.get_global("initial_energy"))
        for k in ("protocol_work", "shadow_work", "total_work"):
            assert abs(g.get_global(k) - o.get_global(k)) <= tol * max(1.0, abs(o.get_global("protocol_work"))) + (slack if k != "protocol_work" else 0.0), (k, n)
    assert g.get_global("step") == 6.0 and g.get_global("lambda_sterics") == o.get_global("lambda_sterics")
    for k in ("initial_energy", "final_energy", "Epert"):
        assert g.get_global(k) == pytest.approx(o.get_global(k), rel=max(tol, 1e-9) * 1e-2 if precision == 0 else 1e-11), k
    assert np.abs(g.get_positions() - o.get_positions()).max() < (1e-9 if precision else 5e-5)
    assert abs(o.get_global("shadow_work")) > 1e-3   # (there is something to compare)
    g.step(3)   # no-op past the end
    assert g.get_global("step") == 6.0
    g.reset()
    assert g.get_global("total_work") == 0.0 and g.get_global("shadow_work") == 0.0 and g.get_global("step") == 0.0


@pytest.mark.parametrize("precision,tol", [(1, 1e-9), (0, 2e-4)])
def test_ghmc_switching_matches_the_oracle(Engine, oracle_mod, tol_box, precision, tol):
    it = switching.NCMCGHMCAlchemicalIntegrator(300.0, None, FUNCS, nsteps=8, timestep=0.002, direction="delete")
    it.setRandomNumberSeed(5)
    g, o = _pair(Engine, oracle_mod, tol_box, it, precision)
    g.step(8); o.step(8)
    assert g.get_global("ntrials") == o.get_global("ntrials") == 9.0
    assert g.get_global("naccept") == o.get_global("naccept")
    assert 0 < o.get_global("naccept") < 9, "the case should see acceptances and rejections"
    assert g.get_global("protocol_work") == pytest.approx(o.get_global("protocol_work"), rel=tol)
    assert g.get_global("total_work") == pytest.approx(o.get_global("total_work"), rel=tol) and g.get_global("shadow_work") == 0.0
    assert np.abs(g.get_positions() - o.get_positions()).max() < (1e-9 if precision else 5e-5)
    assert np.abs(g.get_velocities() - o.get_velocities()).max() < (1e-8 if precision else 5e-3)
    if precision == 1:
        # a second switch after reset(): the Metropolis uniforms carry on in the Philox stream (keyed on a counter reset() does not
        # touch -- keyed on ntrials every switch would meet the same thresholds); engine and oracle stay in step trial by trial
        first = (g.get_global("naccept"), g.get_global("ntrials"))
        g.reset(); o.reset()
        x0 = o.get_positions(); g.set_positions(x0); o.set_positions(x0)
        g.step(8); o.step(8)
        assert g.get_global("ntrials") == o.get_global("ntrials") == first[1]
        assert g.get_global("naccept") == o.get_global("naccept")
        assert g.get_global("total_work") == pytest.approx(o.get_global("total_work"), rel=1e-8)
        assert np.abs(g.get_positions() - o.get_positions()).max() < 1e-8


def test_instantaneous_toggle_and_mirror_accessors(Engine, oracle_mod, tol_box):
    from blues_amd.context import Context
    s, v = tol_box
    it = switching.NCMCVVAlchemicalIntegrator(300.0, s, FUNCS, nsteps=0, direction="delete")
    ctx = Context(s, it, precision="double")
    ctx.setVelocities(v)
    it.step(1)
    o = oracle_mod.Oracle(s, it.to_data(precision=1)); o.set_velocities(v); o.step(1)
    kT = 0.0083144626 * 300.0
    assert it.getTotalWork(ctx) == pytest.approx(o.get_global("total_work") / kT, rel=1e-9)
    assert it.getProtocolWork(ctx) == it.getTotalWork(ctx) and it.getShadowWork(ctx) == 0.0
    assert it.getLogAcceptanceProbability(ctx) == -it.getTotalWork(ctx)
    assert it.getGlobalVariableByName("final_reduced_potential") == pytest.approx(o.get_global("final_energy") / kT, rel=1e-10)
    assert it.getGlobalVariableByName("initial_reduced_potential") == pytest.approx(o.get_global("initial_energy") / kT, rel=1e-10)
    assert it.get_step() == 1.0
    it.reset()
    assert it.getTotalWork(ctx) == 0.0


def test_switching_engines_do_not_join_batches(Engine, tol_box):
    from blues_amd.engine import EngineError, NativeBatch
    s, v = tol_box
    it = switching.NCMCVVAlchemicalIntegrator(300.0, None, FUNCS, nsteps=2)
    e = [Engine(s, it.to_data(replica=r)) for r in range(2)]
    with pytest.raises(EngineError, match="one engine at a time"):
        NativeBatch(e)
