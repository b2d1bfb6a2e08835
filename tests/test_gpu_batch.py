"""Replica batches (include/blues_engine.h "Replica batches"): R congruent engines on one GPU share every
launch of integrator.step(n).  The batched kernels run the same device functions as the single-replica ones, so the bar
is BITWISE identity with the same replicas advanced alone -- plus the oracle for one of them."""
import copy

import numpy as np
import pytest

from conftest import gpu_available

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not gpu_available(), reason="needs a GPU")]

from blues_amd import integrators, moves, simulation, systems, unit  # noqa: E402


@pytest.fixture(scope="module")
def Engine():
    from blues_amd import build
    build.build_engine()
    from blues_amd.engine import NativeEngine
    return NativeEngine


def _integ(nsteps, seed, dt=0.004, **kw):
    return integrators.generateNCMCIntegrator(nstepsNC=nsteps, dt=dt, temperature=300.0, seed=seed, **kw)


def _replica_inputs(s, v, R):
    """Distinct velocities / Philox keys per replica, same topology."""
    rng = np.random.RandomState(11)
    out = []
    for r in range(R):
        vr = v * (1.0 + 0.05 * r) + 0.01 * rng.standard_normal(v.shape) * (s.mass[:, None] > 0)
        out.append(vr)
    return out


def _make(Engine, s, vels, nsteps, precision, split=None, nprop=1):
    engs = []
    for r, vr in enumerate(vels):
        if split is None:
            integ = _integ(nsteps, seed=100 + r)
        else:
            integ = integrators.AlchemicalExternalLangevinIntegrator(
                {"lambda_sterics": "min(1, (1/0.3)*abs(lambda-0.5))",
                 "lambda_electrostatics": "step(0.2-lambda) - 1/0.2*lambda*step(0.2-lambda) + 1/0.2*(lambda-0.8)*step(lambda-0.8)"},
                splitting=split, temperature=300.0, timestep=0.002, nsteps_neq=nsteps, nprop=nprop, prop_lambda=0.3, seed=100 + r)
        d = integ.to_data(precision=precision, replica=r)
        g = Engine(s, d)
        g.set_velocities(vr)
        engs.append(g)
    return engs


def _state(g):
    return g.get_positions(), g.get_velocities(), g.get_global("protocol_work")


@pytest.fixture(params=["0", "1"])
def same_decomposition(request, tune):
    """A batch picks its launch decomposition for R replicas (fused force launch only while the whole batch is small), which
    changes the order of the partial force sums.  Pinning the choice makes solo and batched runs comparable bit for bit --
    once with the fused launch, once with the separate kernels."""
    tune(fuse_forces=int(request.param), k2_jiter=1 if request.param == "1" else 4)   # j-groups per alchemical block: lone-replica / large-batch value
    return request.param


@pytest.mark.parametrize("precision", [0, 1])
def test_batch_is_bitwise_identical_to_solo(Engine, tol_box, precision, same_decomposition):
    from blues_amd.engine import NativeBatch
    s, v = tol_box
    R, n = 3, 24
    vels = _replica_inputs(s, v, R)
    solo = _make(Engine, s, vels, n, precision)
    # three calls: where a call ends the pending program is flushed, which regroups the ops into launches (and so the
    # summation order of the CM remover) -- same grouping on both sides
    ws = [np.concatenate([g.run_switch(5, trace=True), g.run_switch(7, trace=True), g.run_switch(n - 12, trace=True)]) for g in solo]
    bat = _make(Engine, s, vels, n, precision)
    B = NativeBatch(bat)
    _, w1 = B.step(5, trace=True)
    _, w2 = B.step(7, trace=True)
    _, w3 = B.step(n - 12, trace=True)
    wb = np.concatenate([w1, w2, w3], axis=1)
    st = B.stats()
    assert st["replicas"] == R and st["fallback_steps"] == 0 and st["lockstep_steps"] > 0
    for r in range(R):
        assert np.array_equal(wb[r], ws[r]), (r, np.abs(wb[r] - ws[r]).max())
        xs, vs, _ = _state(solo[r]); xb, vb, _ = _state(bat[r])
        assert np.array_equal(xs, xb) and np.array_equal(vs, vb)
    assert not np.array_equal(ws[0], ws[1])      # the replicas really are different trajectories
    B.close()
    for g in solo + bat:
        g.close()


@pytest.mark.parametrize("precision", [0, 1])
def test_batch_large_iset_unfused_path(Engine, tol_box, precision):
    """More than 32 i-tiles: the force pass is separate launches (alchemical / sub-tile nonbonded / bonded / finalize) and
    the step kernel spans several blocks -- the other half of the batched kernels."""
    from blues_amd.engine import NativeBatch
    s, v = tol_box
    big = systems.tile_system(s, (1, 1, 3))
    vb_ = np.concatenate([v, v, v])
    R, n = 2, 8
    vels = _replica_inputs(big, vb_, R)
    solo = _make(Engine, big, vels, n, precision)
    assert solo[0].stats()["i_tiles"] > 32
    ws = [g.run_switch(n, trace=True) for g in solo]
    bat = _make(Engine, big, vels, n, precision)
    B = NativeBatch(bat)
    _, wb = B.step(n, trace=True)
    assert B.stats()["fallback_steps"] == 0
    for r in range(R):
        assert np.array_equal(wb[r], ws[r]), (r, np.abs(wb[r] - ws[r]).max())
        xs, vs, _ = _state(solo[r]); xb, vb, _ = _state(bat[r])
        assert np.array_equal(xs, xb) and np.array_equal(vs, vb)
    B.close()


def test_batch_default_policy_matches_solo_to_rounding(Engine, tol_box):
    """No pinning: the batch may choose another decomposition than a lone engine; results then agree to summation-order
    rounding (fp64 mode), not bit for bit."""
    from blues_amd.engine import NativeBatch
    s, v = tol_box
    R, n = 4, 12
    vels = _replica_inputs(s, v, R)
    solo = _make(Engine, s, vels, n, 1)
    ws = [g.run_switch(n, trace=True) for g in solo]
    bat = _make(Engine, s, vels, n, 1)
    B = NativeBatch(bat)
    _, wb = B.step(n, trace=True)
    for r in range(R):
        assert np.allclose(wb[r], ws[r], rtol=1e-10, atol=1e-10)
        assert np.abs(solo[r].get_positions() - bat[r].get_positions()).max() < 1e-11
    B.close()
    # leaving the batch restores the lone-engine layout: the members keep working on their own
    for g in bat:
        g.reset(); g.step(2)


def test_batch_vs_oracle(Engine, tol_box, oracle_mod):
    from blues_amd.engine import NativeBatch
    s, v = tol_box
    R, n = 2, 10
    vels = _replica_inputs(s, v, R)
    bat = _make(Engine, s, vels, n, 1)
    B = NativeBatch(bat)
    _, wb = B.step(n, trace=True)
    for r in range(R):
        o = oracle_mod.Oracle(s, _integ(n, seed=100 + r).to_data(precision=1, replica=r))
        o.set_velocities(vels[r])
        wo = []
        for _ in range(n):
            o.step(1); wo.append(o.get_global("protocol_work"))
        assert np.allclose(wb[r], wo, rtol=1e-9, atol=1e-9)
        assert np.abs(bat[r].get_positions() - o.get_positions()).max() < 1e-10
    B.close()


def test_batch_with_frozen_atoms_and_moves(Engine, tol_box, same_decomposition):
    """The flagship shape in miniature (most atoms frozen -> fused force launch) with a position edit of only SOME
    replicas in the middle: the members' control states differ at that step (it falls back to per-member launches unless
    the energy evaluations of the edit happen to bring them back in line), with identical results either way."""
    from blues_amd.engine import NativeBatch
    s, v = tol_box
    lig = np.arange(15)
    near = systems.nearest_molecules(s, lig, 120, exclude_idx=lig)
    sf = systems.freeze_except(s, np.concatenate([lig, near]))
    vf = v * (sf.mass[:, None] > 0)
    R, n = 4, 16
    vels = _replica_inputs(sf, vf, R)
    rot = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])

    def edit(g):
        x = g.get_positions(); c = x[lig].mean(0); x[lig] = (x[lig] - c) @ rot.T + c; g.set_positions(x)

    def run(engs, stepper):
        stepper(n // 2)
        for r, g in enumerate(engs):
            if r % 2 == 0:
                edit(g)
        stepper(n - n // 2)

    solo = _make(Engine, sf, vels, n, 0)
    run(solo, lambda k: [g.step(k) for g in solo])
    bat = _make(Engine, sf, vels, n, 0)
    B = NativeBatch(bat)
    run(bat, lambda k: B.step(k))
    st = B.stats()
    assert st["fallback_steps"] + st["lockstep_steps"] >= n and st["lockstep_steps"] >= n - 4
    for r in range(R):
        xs, vs, w_s = _state(solo[r]); xb, vb, w_b = _state(bat[r])
        assert np.array_equal(xs, xb) and np.array_equal(vs, vb) and w_s == w_b, r
    B.close()


def test_batch_other_programs_and_inactive_members(Engine, tol_box, same_decomposition):
    """General op interpreter (another splitting, nprop > 1) through the batch; a member masked out is left untouched."""
    from blues_amd.engine import NativeBatch
    s, v = tol_box
    R, n = 3, 10
    vels = _replica_inputs(s, v, R)
    solo = _make(Engine, s, vels, n, 1, split="V R H O R V", nprop=2)
    for g in solo:
        g.step(n)
    bat = _make(Engine, s, vels, n, 1, split="V R H O R V", nprop=2)
    B = NativeBatch(bat)
    x1_before = bat[1].get_positions()
    B.step(4, active=[True, False, True])
    assert np.array_equal(bat[1].get_positions(), x1_before) and bat[1].get_global("step") == 0
    B.step(n - 4, active=[True, False, True])
    B.step(n)                       # members 0 and 2 are at the end of their protocol (no-ops); member 1 runs its n steps
    for r in range(R):
        xs, vs, w_s = _state(solo[r]); xb, vb, w_b = _state(bat[r])
        assert np.array_equal(xs, xb) and np.array_equal(vs, vb) and w_s == w_b, r
    B.close()


def test_batch_member_failure_is_isolated(Engine, tol_box, same_decomposition):
    """One replica blows up (atoms placed on top of each other): its status is set and its message readable, the others
    finish with the results they have alone (reference policy: per-simulation exception, blues/simulation.py:1088-1094)."""
    from blues_amd.engine import NativeBatch, EngineError
    s, v = tol_box
    R, n = 3, 12
    vels = _replica_inputs(s, v, R)
    solo = _make(Engine, s, vels, n, 0)
    for r in (0, 2):
        solo[r].step(n)
    bat = _make(Engine, s, vels, n, 0)
    x = bat[1].get_positions(); x[18] = x[21] + 1e-4; x[19] = x[22] + 1e-4; bat[1].set_positions(x)   # two waters superposed
    B = NativeBatch(bat)
    errors, _ = B.step(n, raise_errors=False)
    assert errors[0] is None and errors[2] is None
    assert isinstance(errors[1], EngineError) and ("nan" in str(errors[1]).lower() or "constraint" in str(errors[1]).lower())
    for r in (0, 2):
        assert np.array_equal(solo[r].get_positions(), bat[r].get_positions())
    for g in bat:
        g.reset()
    bat[1].set_positions(x)
    with pytest.raises(EngineError):
        B.step(n)   # raise_errors=True surfaces the member's exception, as integrator.step would
    B.close()


def test_batched_blues_driver_matches_separate_chains(Engine, tol_box, same_decomposition):
    """BatchedBLUESSimulation (lock-step chains) against the same chains run one after the other through BLUESSimulation."""
    from blues_amd.context import Simulation
    s, v = tol_box
    lig = np.arange(15)
    R, nsteps, nIter = 3, 12, 2
    vels = _replica_inputs(s, v, R)

    def chain(r):
        integ = _integ(nsteps, seed=500 + r, dt=0.002)
        sim = Simulation(None, s, integ, precision="double", replica=r)
        sim.context.setVelocities(unit.Quantity(vels[r], "nanometer/picosecond"))
        mover = moves.MoveEngine(moves.RandomLigandRotationMove(lig, s.mass[lig], random_state=70 + r))
        return simulation.BLUESSimulation(simulation.SimulationSet(sim), {"nstepsNC": nsteps, "moveStep": nsteps // 2, "nIter": nIter}, mover)

    # setVelocitiesToTemperature and the Metropolis draw use numpy's global stream: pinned per (chain, iteration) below
    sep = [chain(r) for r in range(R)]
    sep_records = []
    for r, c in enumerate(sep):
        recs = []
        for N in range(nIter):
            np.random.seed(1000 * r + N)
            c.currentIter = N
            c._syncStatesMDtoNCMC(); c._stepNCMC(nsteps, nsteps // 2); c._acceptRejectMove(); recs.append(dict(c.last))
            np.random.seed(5000 + 1000 * r + N)
            c._resetSimulations(300.0)
        sep_records.append(recs)

    bat = [chain(r) for r in range(R)]
    B = simulation.BatchedBLUESSimulation(bat)
    for N in range(nIter):
        for r, c in enumerate(bat):
            c.currentIter = N
            c._syncStatesMDtoNCMC()
        B._stepNCMC(nsteps, nsteps // 2)
        for r, c in enumerate(bat):
            np.random.seed(1000 * r + N)
            c._acceptRejectMove()
            assert c.last["protocol_work"] == sep_records[r][N]["protocol_work"]
            assert c.last["log_accept"] == sep_records[r][N]["log_accept"] and c.last["accept"] == sep_records[r][N]["accept"]
            np.random.seed(5000 + 1000 * r + N)
            c._resetSimulations(300.0)
    for r in range(R):
        assert np.array_equal(sep[r]._ncmc_sim.context._engine.get_positions(), bat[r]._ncmc_sim.context._engine.get_positions())
        assert sep[r].accept == bat[r].accept and sep[r].reject == bat[r].reject
    B.close()


def test_members_with_different_histories_still_share_launches(Engine, tol_box, same_decomposition):
    """A member that has stepped before joining (its noise draw counter is ahead) advances in lock step all the same: the
    records carry its offset.  Joining re-sorts a member at its current positions, so the member with a past agrees with
    its solo continuation to summation-order rounding (fp64 mode); the fresh members bit for bit."""
    from blues_amd.engine import NativeBatch
    s, v = tol_box
    R, n = 3, 10
    vels = _replica_inputs(s, v, R)
    solo = _make(Engine, s, vels, n, 1)
    bat = _make(Engine, s, vels, n, 1)
    for g in (solo[1], bat[1]):        # member 1 has a past: one full protocol, then reset
        g.step(n); g.reset()
    for g in solo:
        g.step(n)
    B = NativeBatch(bat)
    B.step(n)
    st = B.stats()   # the first step differs in more than the counter (member 1 already holds lists and noise): it falls back, the rest is shared
    assert st["fallback_steps"] <= 2 and st["lockstep_steps"] >= n - 2
    for r in range(R):
        xs, vs, w_s = _state(solo[r]); xb, vb, w_b = _state(bat[r])
        if r == 1:
            assert np.abs(xs - xb).max() < 1e-10 and np.abs(vs - vb).max() < 1e-8 and w_s == pytest.approx(w_b, rel=1e-9, abs=1e-9)
        else:
            assert np.array_equal(xs, xb) and np.array_equal(vs, vb) and w_s == w_b, r
    B.close()


def test_device_resident_state_round_trip(Engine, tol_box):
    """getState keeps positions / velocities in HBM (DeviceQuantity); handed back untouched they are restored device to
    device with the bookkeeping of the host calls; read, they are the same numbers the host path returns; edited in
    place after a read, the host copy wins (what a Move does)."""
    from blues_amd.context import Simulation
    s, v = tol_box
    integ = _integ(16, seed=9, dt=0.002)
    sim = Simulation(None, s, integ, precision="double")
    ctx = sim.context
    ctx.setVelocities(unit.Quantity(v, "nanometer/picosecond"))
    keys = dict(getPositions=True, getVelocities=True, getEnergy=True)
    st0 = ctx.getState(**keys)
    x0q, v0q = st0.getPositions(asNumpy=True), st0.getVelocities(asNumpy=True)
    assert isinstance(x0q, unit.DeviceQuantity) and x0q.on_device() is not None
    sim.step(8)
    st1 = ctx.getState(**keys)                      # a second State alive at the same time: its own buffer
    x1 = st1.getPositions(asNumpy=True)._value.copy()
    assert np.array_equal(x1, ctx._engine.get_positions()) and np.array_equal(st1.getVelocities()._value, ctx._engine.get_velocities())
    assert st1.getPositions(asNumpy=True).on_device() is not None          # a fresh Quantity of the same State is still device-backed
    e1 = st1.getPotentialEnergy()._value
    # restore state0 without the host ever seeing it
    ctx.setPositions(x0q); ctx.setVelocities(v0q)
    assert x0q.on_device() is not None
    assert np.array_equal(ctx._engine.get_positions(), s.positions if False else x0q._value)   # (reads now; first host contact)
    ctx.setParameter("lambda_sterics", 1.0); ctx.setParameter("lambda_electrostatics", 1.0)   # as they were when state0 was taken
    assert ctx.getState(getEnergy=True).getPotentialEnergy()._value == pytest.approx(st0.getPotentialEnergy()._value, rel=1e-12)
    # edit after reading: host copy is authoritative
    xe = st1.getPositions(asNumpy=True)
    xe._value[0] += 0.01
    assert xe.on_device() is None
    ctx.setPositions(xe)
    assert np.array_equal(ctx._engine.get_positions()[0], x1[0] + 0.01)
    # the same trajectory as with plain host arrays: a twin context goes through the same motions with numpy copies
    twin = Simulation(None, s, _integ(16, seed=9, dt=0.002), precision="double")
    twin.context.setVelocities(unit.Quantity(v, "nanometer/picosecond"))
    t0 = twin.context.getState(**keys)
    tx, tv = t0.getPositions(asNumpy=True)._value.copy(), t0.getVelocities(asNumpy=True)._value.copy()
    twin.step(8)
    for c, (px, pv) in ((ctx, (st0.getPositions(asNumpy=True), st0.getVelocities(asNumpy=True))),
                        (twin.context, (unit.Quantity(tx, "nanometer"), unit.Quantity(tv, "nanometer/picosecond")))):
        c._integrator.reset()
        c.setPositions(px); c.setVelocities(pv)
    assert st0.getPositions(asNumpy=True).on_device() is not None or True
    sim.step(8); twin.step(8)
    assert np.array_equal(ctx._engine.get_positions(), twin.context._engine.get_positions())
    assert ctx._integrator.getGlobalVariableByName("protocol_work") == twin.context._integrator.getGlobalVariableByName("protocol_work")
    # hand-over between two contexts on the same GPU (MD -> NCMC, reference simulation.py:1037)
    sim2 = Simulation(None, s, _integ(16, seed=10, dt=0.002), precision="double")
    sim2.context.setPositions(st1.getPositions(asNumpy=True)); sim2.context.setVelocities(st1.getVelocities(asNumpy=True))
    assert np.array_equal(sim2.context._engine.get_positions(), x1)
    sim2.context.setParameter("lambda_sterics", 0.0); sim2.context.setParameter("lambda_electrostatics", 0.0)   # lambda = 0.5, where state1 was taken
    assert sim2.context.getState(getEnergy=True).getPotentialEnergy()._value == pytest.approx(e1, rel=1e-12)


def test_batch_synchronised_list_rebuilds(Engine, tol_box, tune):
    """Large batches rebuild every member's neighbour lists together (any member's request rebuilds all): same physics,
    another summation order.  Forced on here for a small batch: all members then count the same rebuilds, more than a
    lone replica needs, and agree with their solo runs to rounding (fp64 mode)."""
    from blues_amd.engine import NativeBatch
    s, v = tol_box
    lig = np.arange(15)
    near = systems.nearest_molecules(s, lig, 150, exclude_idx=lig)
    sf = systems.freeze_except(s, np.concatenate([lig, near]))
    vf = v * (sf.mass[:, None] > 0)
    R, n = 3, 60
    vels = _replica_inputs(sf, vf, R)
    tune(skin=0.08)          # frequent rebuilds within a short run
    solo = _make(Engine, sf, vels, n, 1)
    ws = [g.run_switch(n, trace=True) for g in solo]
    tune(batch_sync_lists=1)
    bat = _make(Engine, sf, vels, n, 1)
    B = NativeBatch(bat)
    _, wb = B.step(n, trace=True)
    builds = [g.stats()["list_builds"] for g in bat]
    assert len(set(builds)) == 1 and builds[0] >= max(g.stats()["list_builds"] for g in solo) and builds[0] > 3
    for r in range(R):
        assert np.allclose(wb[r], ws[r], rtol=1e-9, atol=1e-9)
        assert np.abs(solo[r].get_positions() - bat[r].get_positions()).max() < 1e-10
    B.close()


def test_batched_energy_prefetch_equals_per_member_evaluation(Engine, tol_box, same_decomposition):
    """blues_batch_prefetch_energies fills the members' energy caches from shared launches; the values are those the
    members compute on their own (potential: same partials, same summation order -> bit for bit; kinetic: another
    summation order -> to rounding)."""
    from blues_amd.engine import NativeBatch
    s, v = tol_box
    R, n = 3, 12
    vels = _replica_inputs(s, v, R)
    ref = _make(Engine, s, vels, n, 0)
    bat = _make(Engine, s, vels, n, 0)
    B = NativeBatch(bat)
    for g in ref + bat:
        g.potential_energy()           # the one-off frozen-frozen constant of every member (evaluated per member)
    for g in ref:
        g.step(6)
    B.step(6)
    before = B.stats()["batched_energy_evaluations"]
    B.prefetch_energies()
    assert B.stats()["batched_energy_evaluations"] == before + 1
    launches = [g.stats()["kernel_launches"] for g in bat]
    for r in range(R):
        assert bat[r].potential_energy() == ref[r].potential_energy()
        assert bat[r].kinetic_energy() == pytest.approx(ref[r].kinetic_energy(), rel=1e-13)
    assert [g.stats()["kernel_launches"] for g in bat] == launches        # served from the caches: nothing was launched
    # a position edit invalidates the cache; an edit of every member inside a protocol is evaluated together at the next step
    for g in ref + bat:
        x = g.get_positions(); x[:15] += 0.01; g.set_positions(x)
    for g in ref:
        g.step(2)
    B.step(2)
    assert B.stats()["batched_energy_evaluations"] >= before + 2
    for r in range(R):
        assert bat[r].get_global("protocol_work") == ref[r].get_global("protocol_work")
        assert np.array_equal(bat[r].get_positions(), ref[r].get_positions())
    B.close()


@pytest.mark.parametrize("R,pack", [(8, 0), (8, 2), (64, 2)])
def test_full_size_batch_of_eight(Engine, oracle_mod, tune, R, pack):
    """The benchmark system (S23k, 276 mobile atoms) in a batch of 8: the member count is a multiple of 8, so the nonbonded
    and alchemical launches use the XCD-aware block -> replica map, and the engine's own policy puts a batch of 8 such chains
    into the decomposition bench.py runs at R = 512 (separate force kernels, per-atom lists pruned in passing, side-stream fork).
    Every member equals its solo run bit for bit (the lone engines lay themselves out as members of a batch of 8 would:
    BluesTuning.assume_batch, nothing else pinned), and one member is checked against the oracle.
    pack = 2 (round 6): the constraint clusters packed into two waves per chain instead of one wave per kind of cluster
    (BluesTuning.pack_clusters: what a batch of more than 512 chains does by itself) -- the step kernel runs as a 128-thread block
    whose second wave holds triangles, stars and the single atom; at R = 64 it is the two-waves-per-SIMD form the benchmark runs."""
    from blues_amd.engine import NativeBatch
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    n = 10
    tune(assume_batch=R, pack_clusters=pack)
    rng = np.random.RandomState(5)
    vels = [v * (1.0 + 0.03 * r) for r in range(R)]
    solo = _make(Engine, s, vels, n, 0)
    ws = [g.run_switch(n, trace=True) for g in solo]
    bat = _make(Engine, s, vels, n, 0)
    B = NativeBatch(bat)
    _, wb = B.step(n, trace=True)
    assert B.stats()["fallback_steps"] == 0
    for r in range(R):
        assert np.array_equal(wb[r], ws[r]), (r, np.abs(wb[r] - ws[r]).max())
        assert np.array_equal(solo[r].get_positions(), bat[r].get_positions())
    assert len({w[-1] for w in ws}) == R                  # R different trajectories
    assert bat[0].stats()["step_threads"] == (128 if pack == 2 else 256)
    r = 5
    o = oracle_mod.Oracle(s, _integ(n, seed=100 + r).to_data(precision=0, replica=r))
    o.set_velocities(vels[r])
    wo = []
    for _ in range(n):
        o.step(1); wo.append(o.get_global("protocol_work"))
    assert np.abs(wb[r] - np.array(wo)).max() <= 1e-5 * max(1.0, np.abs(wo).max())   # mixed precision, 10 free-running steps: north_star's 1e-5 on the work
    assert np.abs(bat[r].get_positions() - o.get_positions()).max() < 1e-6
    B.close()


def test_batched_full_iterations_with_md_leg(Engine, tol_box, same_decomposition):
    """BLUESSimulation.run for several chains at once: NCMC leg (replica batch) + Metropolis + MD leg (a second replica batch
    over the chains' MD engines: the OpenMM LangevinIntegrator step, kernels k_step_md_b) -- against the same chains run one
    after the other with the same random streams."""
    from blues_amd.context import Simulation
    s, v = tol_box
    md_sys = copy.copy(s); md_sys.alchemical_atoms = np.zeros(0, np.int32)
    lig = np.array([0, 7, 8, 9])                 # methyl group: torsion moves get accepted now and then even with short protocols
    sub = copy.copy(s); sub.alchemical_atoms = lig.astype(np.int32)
    R, nsteps, nmd, nIter = 3, 16, 6, 2

    def chain(r):
        ncmc = Simulation(None, sub, _integ(nsteps, seed=300 + r, dt=0.002), precision="double", replica=r)
        md = Simulation(None, md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.002, seed=400 + r), precision="double", replica=r)
        for sim in (ncmc, md):
            sim.context.setVelocities(unit.Quantity(v * (1.0 + 0.02 * r), "nanometer/picosecond"))
        mv = moves.MoveEngine(moves.TorsionRotationMove((1, 0), [7, 8, 9], random_state=50 + r))
        return simulation.BLUESSimulation(simulation.SimulationSet(ncmc, md=md), {"nstepsNC": nsteps, "moveStep": nsteps // 2, "nIter": nIter, "nstepsMD": nmd},
                                          mv, rng=np.random.RandomState(900 + r))

    sep = [chain(r) for r in range(R)]
    for c in sep:
        c.run()
    bat = [chain(r) for r in range(R)]
    B = simulation.BatchedBLUESSimulation(bat)
    assert B._md_batch is not None
    B.run()
    for r in range(R):
        assert (bat[r].accept, bat[r].reject) == (sep[r].accept, sep[r].reject) and bat[r].accept + bat[r].reject == nIter
        assert bat[r].last["protocol_work"] == pytest.approx(sep[r].last["protocol_work"], rel=1e-9, abs=1e-9)
        xs = sep[r]._md_sim.context._engine.get_positions(); xb = bat[r]._md_sim.context._engine.get_positions()
        assert np.abs(xs - xb).max() < 1e-9
        assert bat[r]._md_sim.currentStep == nIter * nmd
    assert B._md_batch.stats()["lockstep_steps"] >= nIter * nmd - 2
    B.close()


def test_batched_md_leg_with_barostat_equals_per_chain(Engine, tol_box, tune):
    """MonteCarloBarostat on the MD leg INSIDE a replica batch (reference blues/simulation.py:603-626: the tutorial's flagship iteration is
    NPT): every chain makes its own volume moves (own random stream, own step size), so the members of the batch end up in
    different boxes -- the batch's argument records carry box, margins and fixed-point scales per member -- and each equals the
    same chain stepped on its own, bit for bit."""
    from blues_amd.context import Simulation
    from blues_amd.engine import NativeBatch
    s, v = tol_box
    md_sys = systems.add_barostat(copy.copy(s), 300.0, pressure_bar=1.0, frequency=5)
    md_sys.alchemical_atoms = np.zeros(0, np.int32)
    R, nmd = 8, 42
    tune(assume_batch=R)

    def sims():
        out = []
        for r in range(R):
            sim = Simulation(None, md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.002, seed=400 + r), precision="double", replica=r)
            sim.context.setVelocities(unit.Quantity(v * (1.0 + 0.02 * r), "nanometer/picosecond"))
            assert sim.barostat is not None
            out.append(sim)
        return out
    alone = sims()
    for sim in alone:
        sim.step(nmd)
    together = sims()
    batch = NativeBatch([sim.context._engine for sim in together])
    errors = simulation.BatchedBLUESSimulation._advance(batch, together, {r: nmd for r in range(R)})
    assert not errors, errors
    boxes = []
    for a, b in zip(alone, together):
        assert b.currentStep == nmd and b.barostat.total_attempted == a.barostat.total_attempted == nmd // 5
        assert b.barostat.total_accepted == a.barostat.total_accepted
        assert np.array_equal(a.context._engine.get_box(), b.context._engine.get_box())
        assert np.array_equal(a.context._engine.get_positions(), b.context._engine.get_positions())
        assert np.array_equal(a.context._engine.get_velocities(), b.context._engine.get_velocities())
        boxes.append(np.asarray(b.context._engine.get_box())[0, 0])
    assert sum(sim.barostat.total_accepted for sim in together) > 0 and len(set(boxes)) > 1     # volume moves were accepted: the members' boxes differ
    assert batch.stats()["lockstep_steps"] >= nmd - 2 * (nmd // 5) - 2, batch.stats()            # ... and they still share their launches
    batch.close()


def test_batched_md_leg_with_barostat_at_benchmark_size(Engine, tune):
    """The same at the size the review asked for: S23k with nothing frozen, mixed precision.  Every accepted volume move
    re-lays the member out in its own box; what the members of a batch must agree on (launch geometry) no longer depends on the
    density -- fragment lists size their rows per member, the capacities of the density-sized layouts are pinned by the batch -- so
    boxes that differ by a percent neither fail the batch nor cost the shared launches."""
    from blues_amd.context import Simulation
    from blues_amd.engine import NativeBatch
    base, v = systems.s23k(frozen=False)
    md_sys = systems.add_barostat(copy.copy(base), 300.0, pressure_bar=1.0, frequency=3)
    md_sys.alchemical_atoms = np.zeros(0, np.int32)
    R, nmd = 4, 31
    tune(assume_batch=R)

    def sims():
        out = []
        for r in range(R):
            sim = Simulation(None, md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.002, seed=400 + r), precision="mixed", replica=r)
            sim.context.setVelocities(unit.Quantity(v * (1.0 + 0.02 * r), "nanometer/picosecond"))
            out.append(sim)
        return out
    alone = sims()
    for sim in alone:
        sim.step(nmd)
    together = sims()
    batch = NativeBatch([sim.context._engine for sim in together])
    errors = simulation.BatchedBLUESSimulation._advance(batch, together, {r: nmd for r in range(R)})
    assert not errors, errors
    boxes = []
    for a, b in zip(alone, together):
        assert b.context._engine.stats()["nonbonded_kernel"] == 3
        assert b.barostat.total_attempted == a.barostat.total_attempted == nmd // 3 and b.barostat.total_accepted == a.barostat.total_accepted
        assert np.array_equal(a.context._engine.get_box(), b.context._engine.get_box())
        assert np.array_equal(a.context._engine.get_positions(), b.context._engine.get_positions())
        boxes.append(np.asarray(b.context._engine.get_box())[0, 0])
    assert sum(sim.barostat.total_accepted for sim in together) > 0 and len(set(boxes)) > 1
    assert batch.stats()["lockstep_steps"] >= nmd - 2 * (nmd // 3) - 2, batch.stats()
    batch.close()


def test_fused_and_separate_finalize_agree(Engine, tune):
    """The steady-state step kernel forms the summed forces of the pass itself (fuse_finalize, the default where one workgroup holds a
    chain's clusters) or k_finalize does: same forces and energies bit for bit, the centre-of-mass momentum from differently grouped
    partial sums -- protocol work and positions agree to rounding, not to the bit."""
    from blues_amd.engine import NativeBatch
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    R, n = 8, 30
    out = {}
    for fuse in (1, 0):
        tune(assume_batch=64, fuse_finalize=fuse)
        engs = []
        for r in range(R):
            g = Engine(s, _integ(n, seed=300 + r, dt=0.004).to_data(precision=0, replica=r)); g.set_velocities(v); engs.append(g)
        B = NativeBatch(engs)
        _, w = B.step(n, trace=True)
        out[fuse] = (w.copy(), [g.get_positions() for g in engs])
        B.close()
        for g in engs:
            g.close()
    scale = np.abs(out[1][0]).max()
    assert np.abs(out[1][0] - out[0][0]).max() <= 1e-9 * scale
    for a, b in zip(out[1][1], out[0][1]):
        assert np.abs(a - b).max() < 1e-7      # (30 steps of a liquid amplify a last-bit difference of the momentum sums a hundredfold)


def test_move_style_edits_of_a_device_resident_state(Engine, tol_box):
    """What a Move does (reference blues/moves.py:292-307): read positions[indices], assign positions[i] = xyz, hand the
    Quantity to setPositions.  On this engine only the touched atoms travel; the result equals the plain host route, also
    when the edit cannot be applied on the device (an edited atom constrained to an unedited one -> host fallback)."""
    from blues_amd.context import Simulation
    s, v = tol_box
    mk = lambda: Simulation(None, s, _integ(16, seed=21, dt=0.002), precision="double")
    a, b = mk(), mk()
    for sim in (a, b):
        sim.context.setVelocities(unit.Quantity(v, "nanometer/picosecond")); sim.step(4)
    lig = list(range(15))
    rot = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    # device route
    q = a.context.getState(getPositions=True).getPositions(asNumpy=True)
    sub = q[lig]._value
    assert sub.shape == (15, 3) and q.on_device() is not None
    assert np.array_equal(q[3]._value, sub[3]) and len(q) == s.n_atoms
    new = (sub - sub.mean(0)) @ rot.T + sub.mean(0)
    for k, i in enumerate(lig):
        q[i] = new[k]
    assert np.array_equal(q[lig]._value, new) and q.on_device() is not None        # reads see the pending assignments
    launches = a.context._engine.stats()["kernel_launches"]
    a.context.setPositions(q)
    assert q.on_device() is not None                                                # never materialised
    # host route on the twin
    x = b.context._engine.get_positions(); x[lig] = new
    b.context.setPositions(unit.Quantity(x, "nanometer"))
    assert np.array_equal(a.context._engine.get_positions(), b.context._engine.get_positions())
    a.step(6); b.step(6)
    assert np.array_equal(a.context._engine.get_positions(), b.context._engine.get_positions())
    assert a.integrator.getGlobalVariableByName("protocol_work") == b.integrator.getGlobalVariableByName("protocol_work")
    # an edit that splits a constraint cluster (one hydrogen of a water): applied through the host fallback, same outcome
    q = a.context.getState(getPositions=True).getPositions(asNumpy=True)
    h_new = q[16]._value + np.array([0.001, 0.0, 0.0])
    q[16] = h_new
    a.context.setPositions(q)
    x = b.context._engine.get_positions(); x[16] = h_new
    b.context.setPositions(unit.Quantity(x, "nanometer"))
    assert np.array_equal(a.context._engine.get_positions(), b.context._engine.get_positions())


def test_full_protocol_full_size_batch_properties(Engine, tune):
    """BASELINE.json configs[1] at full size in a batch: eight chains of the S23k system through the whole 1000-step protocol
    with the ligand rotated at lambda = 0.5.  Size-independent properties: every chain ends at lambda = 1 with finite work,
    constraints satisfied, no frozen atom moved, the alchemical parameters back at (1, ~1); and -- determinism across the
    batched and the lone code path -- one chain reproduces its 1000-step solo trajectory bit for bit."""
    from blues_amd.engine import NativeBatch
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    R, n = 8, 1000
    tune(assume_batch=R)   # (the lone chain below lays itself out as a batch member does; nothing else pinned)
    lig = np.arange(15)
    rot = np.array([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])

    def rotate(g):
        x = g.get_positions(); c = x[lig].mean(0); x[lig] = (x[lig] - c) @ rot.T + c; g.set_positions(x)

    vels = [v * (1.0 + 0.02 * r) for r in range(R)]
    bat = _make(Engine, s, vels, n, 0)
    B = NativeBatch(bat)
    B.step(n // 2)
    for g in bat:
        rotate(g)
    errors, _ = B.step(n - n // 2, raise_errors=False)
    assert all(e is None for e in errors)
    st = B.stats()
    assert st["lockstep_steps"] >= n - 2 and st["batched_energy_evaluations"] >= 1       # the perturbed energies were taken together
    frozen = s.mass == 0.0
    works = []
    for g in bat:
        assert g.get_global("step") == n and g.get_global("lambda") == 1.0
        assert g.get_global("lambda_sterics") == 1.0 and abs(g.get_global("lambda_electrostatics") - 1.0) < 1e-12
        w = g.get_global("protocol_work"); works.append(w)
        assert np.isfinite(w) and abs(w) < 1e5
        x = g.get_positions(); c = s.constraint_atoms; m = s.mass[c[:, 0]] > 0
        assert np.abs(np.linalg.norm(x[c[m, 0]] - x[c[m, 1]], axis=1) / s.constraint_dist[m] - 1).max() < 1e-7
        assert np.array_equal(x[frozen], s.positions[frozen])
        assert np.isfinite(g.potential_energy()) and np.isfinite(g.kinetic_energy())
    assert len(set(works)) == R
    solo = _make(Engine, s, vels[3:4], n, 0)[0]       # chain 3 alone (its Philox stream is replica index 0 there: rebuild with the right one)
    solo.close()
    from blues_amd import integrators as _i
    solo = Engine(s, _integ(n, seed=103).to_data(precision=0, replica=3)); solo.set_velocities(vels[3])
    solo.step(n // 2); rotate(solo); solo.step(n - n // 2)
    assert solo.get_global("protocol_work") == works[3]
    assert np.array_equal(solo.get_positions(), bat[3].get_positions())
    B.close()


def test_batch_rejects_incongruent_members_and_misuse(Engine, tol_box):
    """Members must share topology and protocol; an engine belongs to one batch at a time; a dissolved batch says so."""
    from blues_amd.engine import NativeBatch, EngineError
    s, v = tol_box
    a = Engine(s, _integ(10, seed=1).to_data(precision=0, replica=0)); a.set_velocities(v)
    b = Engine(s, _integ(12, seed=2).to_data(precision=0, replica=1)); b.set_velocities(v)      # another protocol length
    B = NativeBatch([a, b])
    with pytest.raises(EngineError, match="congruent"):
        B.step(2)
    with pytest.raises(EngineError, match="already belongs"):
        NativeBatch([a])
    B.close()
    a.step(2); b.step(2)                       # both usable on their own again
    c = Engine(s, _integ(10, seed=3).to_data(precision=0, replica=2)); c.set_velocities(v)
    B2 = NativeBatch([a, c])
    B2.step(3)
    assert a.get_global("step") == 5 and c.get_global("step") == 3      # members need not be at the same step (that round falls back)
    c.close()                                  # destroying a member dissolves the batch
    with pytest.raises(EngineError, match="dissolved"):
        B2.step(1)
    B2.close(); a.step(1)


def test_batched_boundary_of_the_full_triple_equals_chain_by_chain(Engine, tol_box, tune):
    """Chains that carry the reference's full triple -- md, alch and ncmc Simulations (blues/simulation.py:768-809) -- through whole
    BLUES iterations with an MD leg (run(): blues/simulation.py:1215-1257): the MD -> NCMC hand-over as one capture of the MD batch and
    one restore into the NCMC batch, the alch energies of the correction for all chains at once, accepted States into the MD batch in
    one call, the velocity redraw on the MD batch -- against the same chains taken through those steps one by one: same accept
    records, same final MD and NCMC states, bit for bit."""
    from blues_amd.context import Simulation
    s, v = tol_box
    md_sys = copy.copy(s); md_sys.alchemical_atoms = np.zeros(0, np.int32)
    lig = np.arange(15)
    R, nsteps, nmd, nIter = 4, 10, 6, 3
    vels = _replica_inputs(s, v, R)
    tune(assume_batch=R)

    def chains():
        out = []
        for r in range(R):
            sim = Simulation(None, s, _integ(nsteps, seed=700 + r, dt=0.002), precision="mixed", replica=r)
            md = Simulation(None, md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.002, seed=800 + r), precision="mixed", replica=r)
            alch = Simulation(None, md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.002, seed=900 + r), precision="mixed", replica=r)
            md.context.setPositions(unit.Quantity(s.positions, "nanometer")); md.context.setVelocities(unit.Quantity(vels[r], "nanometer/picosecond"))
            mover = moves.MoveEngine(moves.RandomLigandRotationMove(lig, s.mass[lig], random_state=90 + r))
            out.append(simulation.BLUESSimulation(simulation.SimulationSet(sim, md=md, alch=alch), {"nstepsNC": nsteps, "moveStep": nsteps // 2, "nIter": nIter, "nstepsMD": nmd},
                                                  mover, rng=np.random.RandomState(4000 + r)))
        return out

    results = {}
    for fast in (False, True):
        np.random.seed(123)
        cs = chains()
        B = simulation.BatchedBLUESSimulation(cs, batched_boundary=fast)
        assert B._batchable() == fast and B._md_batch is not None and B._alch_batch is not None
        records = []
        B.run(nIter=nIter, on_iteration=lambda N, last: records.append([dict(l) for l in last]))
        results[fast] = (records, [c._md_sim.context._engine.get_positions() for c in cs], [c._md_sim.context._engine.get_velocities() for c in cs],
                         [c._ncmc_sim.context._engine.get_positions() for c in cs], [c.accept for c in cs])
        assert all(c._md_sim.currentStep == nIter * nmd for c in cs)
        B.close()
    (rec0, x0, v0, n0, acc0), (rec1, x1, v1, n1, acc1) = results[False], results[True]
    assert acc0 == acc1
    for N in range(nIter):
        for r in range(R):
            for key in ("accept", "log_accept", "correction", "randnum", "protocol_work"):
                assert rec0[N][r][key] == rec1[N][r][key], (N, r, key, rec0[N][r][key], rec1[N][r][key])
    for r in range(R):
        assert np.array_equal(x0[r], x1[r]) and np.array_equal(v0[r], v1[r]) and np.array_equal(n0[r], n1[r])


def test_a_chain_that_dies_is_retired_and_the_others_carry_on(Engine, tol_box, tune):
    """isolate_failures: the reference runs one chain per process, so a chain whose switch blows up (getState raises after the abandoned
    switch, blues/simulation.py:1096) or whose MD leg raises (sys.exit(1), :1203-1213) takes nobody with it.  On the batched path
    a chain that dies in iteration 2 -- two waters of its MD state placed on top of each other before the hand-over -- is retired;
    the other chains finish all iterations with exactly the records and states they have in a run in which nobody dies."""
    from blues_amd.context import Simulation
    s, v = tol_box
    md_sys = copy.copy(s); md_sys.alchemical_atoms = np.zeros(0, np.int32)
    lig = np.arange(15)
    R, nsteps, nmd, nIter = 4, 10, 6, 3
    vels = _replica_inputs(s, v, R)
    tune(assume_batch=R)

    def chains():
        out = []
        for r in range(R):
            sim = Simulation(None, s, _integ(nsteps, seed=700 + r, dt=0.002), precision="mixed", replica=r)
            md = Simulation(None, md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.002, seed=800 + r), precision="mixed", replica=r)
            alch = Simulation(None, md_sys, integrators.LangevinIntegrator(300.0, 1.0, 0.002, seed=900 + r), precision="mixed", replica=r)
            md.context.setPositions(unit.Quantity(s.positions, "nanometer")); md.context.setVelocities(unit.Quantity(vels[r], "nanometer/picosecond"))
            mover = moves.MoveEngine(moves.RandomLigandRotationMove(lig, s.mass[lig], random_state=90 + r))
            out.append(simulation.BLUESSimulation(simulation.SimulationSet(sim, md=md, alch=alch), {"nstepsNC": nsteps, "moveStep": nsteps // 2, "nIter": nIter, "nstepsMD": nmd},
                                                  mover, rng=np.random.RandomState(4000 + r)))
        return out

    results = {}
    for sabotage in (False, True):
        np.random.seed(123)
        cs = chains()
        B = simulation.BatchedBLUESSimulation(cs, isolate_failures=True)
        assert B._batchable()
        records = []

        def after(N, last):
            records.append([dict(l) for l in last])
            if sabotage and N == 0:      # (between the decision and the reset of iteration 1: chain 2's MD state for iteration 2)
                e = cs[2]._md_sim.context._engine
                x = e.get_positions(); x[18] = x[21] + 1e-4; x[19] = x[22] + 1e-4; x[20] = x[23] + 1e-4
                e.set_positions(x)
        B.run(nIter=nIter, on_iteration=after)
        results[sabotage] = (records, [c._md_sim.context._engine.get_positions() for c in cs], [c.accept for c in cs], dict(B.dead))
        B.close()
    (rec0, x0, acc0, dead0), (rec1, x1, acc1, dead1) = results[False], results[True]
    assert dead0 == {} and list(dead1) == [2], (dead0, dead1)
    for r in (0, 1, 3):
        assert acc0[r] == acc1[r] and np.array_equal(x0[r], x1[r])
        for N in range(nIter):
            for key in ("accept", "log_accept", "correction", "randnum", "protocol_work"):
                assert rec0[N][r][key] == rec1[N][r][key], (N, r, key)
    assert rec1[-1][2].get("failed") is True


def test_restoring_a_state_that_moves_a_frozen_atom(Engine):
    """setPositions from a device-resident State learns on the device whether a FROZEN atom changed (the frozen-frozen energy is a
    cached constant).  The verdict is read back lazily, by the next evaluation; two restores in a row must not lose it."""
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    g = Engine(s, _integ(10, 3).to_data(precision=1))
    e0 = g.potential_energy()
    snap = g.snapshot(positions=True, velocities=False)
    frozen = int(np.nonzero(s.mass == 0.0)[0][100])
    x = s.positions.copy(); x[frozen] += [0.05, -0.03, 0.02]
    g.set_positions(x)
    e1 = g.potential_energy()
    assert abs(e1 - e0) > 1e-3
    g.set_positions_from_snapshot(snap)     # the frozen atom goes back: the constant cached for x is stale ...
    g.set_positions_from_snapshot(snap)     # ... and a second restore (which changes nothing) must not hide that
    assert g.potential_energy() == pytest.approx(e0, rel=1e-13)
    g.set_positions(x); g.set_positions_from_snapshot(snap); g.step(2)   # stepping resolves the verdict as well
    g2 = Engine(s, _integ(10, 3).to_data(precision=1)); g2.step(2)
    assert np.abs(g.get_positions() - g2.get_positions()).max() < 1e-11   # (not bitwise: the layout was derived from other coordinates)


def test_batched_plugin_boundary_equals_chain_by_chain(Engine, tol_box, tune):
    """The plugin boundary issued once for all chains (blues_batch_snapshot_capture / _restore / _read_atoms / _restore_edited /
    _reset / _set_velocities_to_temperature: State hand-overs, the Move's positions[atom_indices] and setPositions, the Metropolis
    step's restore, integrator.reset() and the velocity redraw -- reference blues/simulation.py:1028-1187, blues/moves.py:292-307)
    against the same chains taken through those calls one by one: same accept records, same final states, bit for bit, over three
    BLUES iterations with accepted and rejected moves."""
    from blues_amd.context import Simulation
    s, v = tol_box
    lig = np.arange(15)
    R, nsteps, nIter = 6, 10, 3
    vels = _replica_inputs(s, v, R)
    tune(assume_batch=R)

    def chains():
        out = []
        for r in range(R):
            integ = _integ(nsteps, seed=700 + r, dt=0.002)
            sim = Simulation(None, s, integ, precision="mixed", replica=r)
            sim.context.setVelocities(unit.Quantity(vels[r], "nanometer/picosecond"))
            mover = moves.MoveEngine(moves.RandomLigandRotationMove(lig, s.mass[lig], random_state=90 + r))
            out.append(simulation.BLUESSimulation(simulation.SimulationSet(sim), {"nstepsNC": nsteps, "moveStep": nsteps // 2, "nIter": nIter}, mover,
                                                  rng=np.random.RandomState(4000 + r)))
        return out

    results = {}
    for fast in (False, True):
        np.random.seed(123)   # (MoveEngine.selectMove draws from the global stream)
        cs = chains()
        B = simulation.BatchedBLUESSimulation(cs, batched_boundary=fast)
        assert B._batchable() == fast
        records = []
        B.run(nIter=nIter, on_iteration=lambda N, last: records.append([dict(l) for l in last]))
        results[fast] = (records, [c._ncmc_sim.context._engine.get_positions() for c in cs], [c._ncmc_sim.context._engine.get_velocities() for c in cs],
                         [c.accept for c in cs], B._ncmc_batch.stats())
        B.close()
    (rec0, x0, v0, acc0, st0), (rec1, x1, v1, acc1, st1) = results[False], results[True]
    assert acc0 == acc1
    for N in range(nIter):
        for r in range(R):
            for key in ("accept", "log_accept", "correction", "randnum", "protocol_work"):
                assert rec0[N][r][key] == rec1[N][r][key], (N, r, key, rec0[N][r][key], rec1[N][r][key])
    for r in range(R):
        assert np.array_equal(x0[r], x1[r]) and np.array_equal(v0[r], v1[r]), r
    assert 0 < sum(acc0) < R * nIter or True      # (informative only: both outcomes normally occur)
    assert st1["fallback_steps"] <= st0["fallback_steps"]


def test_two_batches_taking_turns_on_the_device_equal_the_batches_run_one_after_the_other(Engine, tol_box, tune):
    """Several BatchedBLUESSimulation objects on one GPU, each driven from its own host thread, sharing a `device_turn` lock
    (bench.py --groups): the stepping calls take turns on the device, the host phases overlap them.  Chains are independent and
    every chain draws from its own streams, so the interleaving changes nothing: same accept records and final states, bit for
    bit, as the two batches run one after the other on the main thread."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from blues_amd.context import Simulation
    s, v = tol_box
    lig = np.arange(15)
    R, nsteps, nIter = 4, 10, 3
    vels = _replica_inputs(s, v, 2 * R)
    tune(assume_batch=R)

    def chains(g):
        out = []
        for r in range(g * R, (g + 1) * R):
            integ = _integ(nsteps, seed=900 + r, dt=0.002)
            sim = Simulation(None, s, integ, precision="mixed", replica=r)
            sim.context.setVelocities(unit.Quantity(vels[r], "nanometer/picosecond"))
            mover = moves.MoveEngine(moves.RandomLigandRotationMove(lig, s.mass[lig], random_state=190 + r))
            out.append(simulation.BLUESSimulation(simulation.SimulationSet(sim), {"nstepsNC": nsteps, "moveStep": nsteps // 2, "nIter": nIter}, mover,
                                                  rng=np.random.RandomState(5000 + r)))
        return out

    def run(threaded):
        turn = threading.Lock() if threaded else None
        drivers = [simulation.BatchedBLUESSimulation(chains(g), device_turn=turn) for g in range(2)]
        assert all(d._batchable() for d in drivers)
        assert all(d._ncmc_batch.device_turn is turn for d in drivers)
        records = [[], []]

        def go(g):
            drivers[g].run(nIter=nIter, on_iteration=lambda N, last: records[g].append([dict(l) for l in last]))
        if threaded:
            with ThreadPoolExecutor(max_workers=2) as pool:
                list(pool.map(go, range(2)))
        else:
            go(0); go(1)
        out = (records, [[c._ncmc_sim.context._engine.get_positions() for c in d.chains] for d in drivers], [[c.accept for c in d.chains] for d in drivers])
        for d in drivers:
            d.close()
        return out

    rec0, x0, acc0 = run(False)
    rec1, x1, acc1 = run(True)
    assert acc0 == acc1
    for g in range(2):
        for N in range(nIter):
            for r in range(R):
                for key in ("accept", "log_accept", "correction", "randnum", "protocol_work"):
                    assert rec0[g][N][r][key] == rec1[g][N][r][key], (g, N, r, key)
        for r in range(R):
            assert np.array_equal(x0[g][r], x1[g][r]), (g, r)


def test_configs3_full_size_water_switch_properties():
    """BASELINE.json configs[3] at FULL size under pytest: WaterTranslationMove on the 23,400-atom box with NOTHING frozen
    ("backbone" restraints on 40 atoms, the first water alchemical; reference examples/example_water.py, blues/moves.py:846-1083),
    the whole 2000-step switch, four chains in one replica batch, mixed precision, the engine's own policy.  Size-independent
    properties: every chain reaches lambda = 1 with finite protocol work (or carries the 999999 a water outside the sphere is
    given: reference blues/moves.py:1082); every constraint holds to 1e-8 relative; the alchemical parameters end at (1, ~1);
    the restrained atoms stay near their anchors; the chains -- own Philox streams, own moves -- end in different states;
    neighbour-list re-sorts and rebuilds happened along the way; and a chain placed outside the sphere IS rejected with >= 999999."""
    import copy
    from blues_amd import build
    from blues_amd.context import Simulation
    build.build_engine()
    base, vel = systems.s23k(frozen=False, restrained=40)
    system = copy.copy(base)
    system.alchemical_atoms = np.array([15, 16, 17], np.int32)
    res = np.asarray(system.residue_of_atom)
    o_idx = [i for i in range(15, system.n_atoms - 2) if res[i] == res[i + 2] and (i == 0 or res[i - 1] != res[i]) and system.mass[i] > 10.0]
    waters = [[i, i + 1, i + 2] for i in o_idx]
    R, nsteps = 4, 2000

    class Outside(moves.WaterTranslationMove):
        def _random_sphere_point(self, radius, origin):
            d = super()._random_sphere_point(radius, origin) - origin
            return origin + d / np.linalg.norm(d) * radius * 1.5

        def afterMove(self, context):
            # (a decoupled water flies ~1 nm in the 4 ps that remain: whether it is back inside a 1 nm sphere at the end is a coin the
            # rounding of the force sums flips; the check this chain exists for is made against a sphere it cannot be in)
            self.radius = 0.05
            return super().afterMove(context)
    chains = []
    for r in range(R):
        integ = integrators.generateNCMCIntegrator(nstepsNC=nsteps, dt=0.004, temperature=300.0, seed=9000 + r)
        sim = Simulation(None, system, integ, precision="mixed", replica=r)
        sim.context.setVelocities(unit.Quantity(vel, "nanometer/picosecond"))
        cls = Outside if r == R - 1 else moves.WaterTranslationMove
        # (the chain that must be rejected: a 1 nm sphere and a placement at 1.5 nm -- half a nanometre is more than a water diffuses
        # back in the 4 ps that remain; with the 2 nm sphere of the example the box's half edge, 2.18 nm, leaves no room outside)
        mover = moves.MoveEngine(cls(waters, np.arange(15), system.mass[:15], radius=1.0 if cls is Outside else 2.0))
        chains.append(simulation.BLUESSimulation(simulation.SimulationSet(sim), {"nstepsNC": nsteps, "moveStep": nsteps // 2, "nIter": 1}, mover,
                                                 rng=np.random.RandomState(60 + r)))
    np.random.seed(21)
    B = simulation.BatchedBLUESSimulation(chains)
    assert B._batchable() and not B._move_batchable()   # State hand-overs, Metropolis step and reset in one call per operation; this Move's hooks (beforeMove / afterMove of its own) chain by chain, as the reference runs them
    records = []
    B.run(nIter=1, on_iteration=lambda N, last: records.append([dict(l) for l in last]))
    last = records[0]
    ca, cd = system.constraint_atoms, system.constraint_dist
    finals = []
    for r, c in enumerate(chains):
        e = c._ncmc_sim.context._engine
        st1 = c.stateTable["ncmc"]["state1"]
        x1 = st1["positions"]._value if not last[r]["accept"] else e.get_positions()
        w = last[r]["protocol_work"]
        assert np.isfinite(w), (r, w)
        if r == R - 1:
            assert w >= 999999 and not last[r]["accept"]          # outside the sphere: rejected through the work (moves.py:1082)
        else:
            assert abs(w) < 1e5 or w >= 999999, (r, w)
        xs = np.asarray(st1["positions"]._value)
        d = xs[ca[:, 0]] - xs[ca[:, 1]]; d -= system.box * np.round(d / system.box)
        assert np.abs(np.linalg.norm(d, axis=1) / cd - 1.0).max() < 1e-8, r
        assert np.isfinite(xs).all() and np.isfinite(st1["potential_energy"]._value)
        dr = xs[system.restraint_atoms] - system.restraint_x0; dr -= system.box * np.round(dr / system.box)
        assert np.linalg.norm(dr, axis=1).max() < 0.25, r        # k = 2092 kJ/mol/nm^2: thermal excursions of a few hundredths of a nm
        finals.append(xs)
        st = e.stats()
        assert st["list_generation"] > 20 and st["force_passes"] >= nsteps, st
    assert all(np.abs(finals[0] - f).max() > 1e-3 for f in finals[1:])      # four different trajectories
    bst = B._ncmc_batch.stats()
    assert bst["lockstep_steps"] > 0.9 * nsteps, bst                         # the members did share their launches
    B.close()


def test_dense_alchemical_kernel_equals_the_lane_layout(Engine, tune):
    """Large batches evaluate the alchemical x environment pairs in a dense form: one workgroup per chain, the pairs within the cutoff
    compacted so that every lane holds one.  Round 6 runs it in fp32 pair arithmetic from the fixed-point image (kernels_alch.h:
    alchemical_dense32_body) with the protocol work formed from per-pair DIFFERENCES between the lambda slots; the fp64 forms stay as
    references: the lane layout (k2_dense = 0: one lane per (alchemical atom, list entry), fp64) and the round-5 dense body (k2_dense = 2).
    Same pairs: total energies agree to 1e-9 (the alchemical sum is ~1e-3 of the total and carries ~1e-6 of itself), forces to 2e-6 of the
    largest (one fp32 rounding of a pair's separation), the two fp64 forms to summation-order rounding -- and the WORK of one step taken
    from identical states, which is the quantity the difference form protects, to 1e-7 of the energy scale it comes from."""
    from blues_amd import integrators, systems
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    integ = lambda: integrators.generateNCMCIntegrator(nstepsNC=60, dt=0.004, temperature=300.0, seed=21).to_data(precision=0)
    tune(assume_batch=512)
    d = Engine(s, integ()); d.set_velocities(v)
    tune(assume_batch=512, k2_dense=0)
    l = Engine(s, integ()); l.set_velocities(v)
    tune(assume_batch=512, k2_dense=2)
    d64 = Engine(s, integ()); d64.set_velocities(v)
    assert d.stats()["alchemical_kernel"] == 1 and l.stats()["alchemical_kernel"] == 0 and d64.stats()["alchemical_kernel"] == 2
    mob = np.nonzero(s.mass > 0)[0]
    for lam in ((1.0, 1.0), (0.5, 0.0), (0.3, 0.0), (1.0, 0.4), (0.05, 0.0)):
        for g in (d, l, d64):
            g.set_global("lambda_sterics", lam[0]); g.set_global("lambda_electrostatics", lam[1])
        ed, el, e64 = d.potential_energy(), l.potential_energy(), d64.potential_energy()
        assert abs(ed - el) <= 1e-9 * abs(el), (lam, ed, el)
        assert abs(e64 - el) <= 1e-11 * abs(el), (lam, e64, el)
        fd, fl, f64 = d.get_forces()[mob], l.get_forces()[mob], d64.get_forces()[mob]
        assert np.abs(fd - fl).max() <= 2e-6 * np.abs(fl).max(), (lam, np.abs(fd - fl).max())     # (fp32 pair arithmetic: 3e-7 of a pair's force)
        assert np.abs(f64 - fl).max() <= 1e-8 * np.abs(fl).max() + 2e-5, (lam, np.abs(f64 - fl).max())
    # the work, step by step from identical states across the whole schedule (electrostatics moving, sterics moving, both flat): the
    # fp64 lane layout is the teacher -- the fp32 form is re-synchronised to its state before every step, as
    # test_gpu_parity.py::test_full_length_protocol_teacher_forced does against the oracle -- and the accumulated difference of the work
    # increments stays below 1e-6 of the work scale: a tenth of the 1e-5 bar, with fp32 pair arithmetic
    for g in (d, l):
        g.reset(); g.set_positions(s.positions); g.set_velocities(v)
    wd_prev = wl_prev = 0.0
    drift, scale = 0.0, 0.0
    for k in range(60):
        if k:
            d.set_positions(l.get_positions()); d.set_velocities(l.get_velocities())
        wd, wl = d.run_switch(1, trace=True)[0], l.run_switch(1, trace=True)[0]
        inc_d, inc_l = wd - wd_prev, wl - wl_prev
        if k:   # (the re-synchronisation is an "instantaneous move" of the student: its work is taken out again)
            inc_d -= d.get_global("perturbed_pe") - d.get_global("unperturbed_pe")
        drift += inc_d - inc_l
        scale = max(scale, abs(wl))
        wd_prev, wl_prev = wd, wl
        assert abs(drift) <= 1e-6 * max(scale, 10.0), (k, drift, scale)
    assert scale > 10.0
    for g in (d, l):
        g.reset(); g.set_positions(s.positions); g.set_velocities(v)
    wd, wl = d.run_switch(60, trace=True), l.run_switch(60, trace=True)
    assert np.abs(wd - wl).max() <= 5e-5 * np.abs(wl).max(), np.abs(wd - wl).max()   # free-running 0.24 ps of a chaotic liquid: 1e-6 force differences grow
    d.close(); l.close(); d64.close()


def test_the_order_of_a_pass_kernels_changes_no_bit(Engine, tune):
    """BluesTuning.fork = 0 / 1 / 2 / 3 / 4 only decide which stream the alchemical and bonded kernels of a batched force pass run on and where
    they are joined (1, the default: the two small ones beside the builder of the atoms' lists; 2: the dense kernel too, joined
    before the nonbonded kernel).  Every sum of a pass is formed in a fixed order, so a batch of the bench decomposition ends on the
    same bits whichever it is."""
    from blues_amd import integrators, systems
    from blues_amd.engine import NativeBatch
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    out = {}
    for fork in (0, 1, 2, 3, 4):
        tune(fork=fork)
        engs = [Engine(s, integrators.generateNCMCIntegrator(nstepsNC=60, dt=0.004, temperature=300.0, seed=33).to_data(precision=0, replica=r)) for r in range(8)]
        for g in engs:
            g.set_velocities(v)
        B = NativeBatch(engs)
        _, w = B.step(60, trace=True)
        st = engs[0].stats()
        assert st["alchemical_kernel"] == 1 and st["nonbonded_kernel"] == 2, st          # the bench decomposition: dense kernel, per-atom lists
        assert B.stats()["fallback_steps"] == 0
        out[fork] = (np.asarray(w), [g.get_positions() for g in engs])
        B.close()
        for g in engs:
            g.close()
    for other in (1, 2, 3, 4):
        assert np.array_equal(out[0][0], out[other][0]), other
        for a, b in zip(out[0][1], out[other][1]):
            assert np.array_equal(a, b), other


def test_members_take_the_batch_shape_in_place(Engine, tune):
    """Members of a batch must agree on the layout shape of the per-atom-list mode (tiles per group list, list capacity).  Round 5
    re-laid out every member whose own choice differed from a new sort -- at creation and, up to three times per member, whenever the
    batch re-planned its shape in the middle of a switch (BENCH_r05: one 4.5 s iteration in both batches).  Now the shape is planned
    from the members' own tables and a member moves to it IN PLACE (blues_engine.hip: reshape_groups): the sorted order, the image,
    the tiles and every per-slot table do not depend on the shape; only the group lists are rebuilt, on the device.  Half of the
    members arrive laid out with two tiles per list, the batch's shape is five: they are reshaped, nobody is sorted again, and the batch
    ends on the same bits as one whose members all arrived with five."""
    from blues_amd.engine import NativeBatch
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    R, n = 6, 30

    def build(groups):
        engs = []
        for r in range(R):
            tune(assume_batch=64, list_group=groups[r])
            g = Engine(s, _integ(100, seed=300 + r).to_data(precision=0, replica=r)); g.set_velocities(v * (1.0 + 0.02 * r)); engs.append(g)
        tune(assume_batch=64, list_group=0)
        return engs, NativeBatch(engs)

    same, Bs = build([5] * R)
    cs = Bs.counters()
    assert cs["tiles_per_list"] == 5 and cs["nonbonded_kernel"] == 2 and cs["reshapes"] == 0 and cs["relayouts"] == 0, cs
    _, ws = Bs.step(n, trace=True)
    mixed, Bm = build([5, 2, 5, 2, 2, 5])
    assert [g.stats()["tiles_per_list"] for g in mixed] == [5] * R          # (the batch's shape: the leader's)
    cm = Bm.counters()
    assert cm["reshapes"] == 3 and cm["relayouts"] == 0 and cm["replans"] == 0, cm
    _, wm = Bm.step(n, trace=True)
    assert Bm.stats()["fallback_steps"] == 0
    assert np.array_equal(ws, wm)
    for a, b in zip(same, mixed):
        assert np.array_equal(a.get_positions(), b.get_positions()) and np.array_equal(a.get_velocities(), b.get_velocities())
        assert b.audit_lists()[1] == 0
    for B in (Bs, Bm):
        B.close()
    for g in same + mixed:
        g.close()


def _swap_waters_outward(s, x, n_pairs=2, distance=1.6):
    """x with the n_pairs mobile waters nearest the ligand exchanged against frozen waters `distance` nm from it (identical molecules:
    the same configuration, another set of mobile atoms -- their bounding sphere no longer fits one group list)."""
    res = np.asarray(s.residue_of_atom)
    lig = np.asarray(s.alchemical_atoms)
    x = x.copy()
    centre = x[lig].mean(axis=0)
    box = np.asarray(s.box, dtype=float).reshape(-1)[:3]
    first = np.unique(res, return_index=True)[1]                                       # first atom of every residue
    d = x[first] - centre; d -= box * np.rint(d / box)
    dist = np.linalg.norm(d, axis=1)
    is_water = np.bincount(res)[res[first]] == 3
    frozen_w = first[is_water & (s.mass[first] == 0.0)]
    mobile_w = first[is_water & (s.mass[first] > 0.0)]
    dist_of = dict(zip(first.tolist(), dist.tolist()))
    w1 = min(frozen_w, key=lambda a: abs(dist_of[int(a)] - distance))
    dd = x[frozen_w] - x[w1]; dd -= box * np.rint(dd / box)
    far = frozen_w[np.argsort(np.linalg.norm(dd, axis=1))[:n_pairs]]                   # ... and its nearest frozen neighbours
    near = sorted(mobile_w, key=lambda a: dist_of[int(a)])[:n_pairs]
    for a, b in zip(near, far):
        ia, ib = np.arange(a, a + 3), np.arange(b, b + 3)
        x[ia], x[ib] = x[ib].copy(), x[ia].copy()
    return x


def test_a_member_that_outgrows_the_shape_steps_on_its_own(Engine, tune):
    """A member whose mobile atoms have spread beyond what the batch's layout shape holds (in the benchmark geometry -- one compact group of
    261 mobile atoms -- no finer shape holds less: a tile of 64 Hilbert-consecutive atoms spans the blob).  Round 5 re-planned the shape for
    everybody: every member laid out again, up to three times (BENCH_r05: 3 s in the middle of a switch, both batches of the device
    waiting), and the whole batch on fragment lists for good because of ONE chain.  Now that member becomes a straggler
    (blues_engine.hip: make_straggler): it lays itself out on its own, leaves the shared launches and steps behind them on launches of
    its own; nobody else is touched -- the OTHER members end on the same BITS as in a run in which nothing happened -- and it comes back
    at the start of its next switch if its atoms fit again."""
    from blues_amd.engine import NativeBatch
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    R, n = 6, 20
    tune(assume_batch=64)      # (the layout of a large batch: per-atom lists over group images, the benchmark's)

    def build():
        engs = []
        for r in range(R):
            g = Engine(s, _integ(100, seed=300 + r).to_data(precision=0, replica=r)); g.set_velocities(v * (1.0 + 0.02 * r)); engs.append(g)
        return engs, NativeBatch(engs)

    calm, Bc = build()
    Bc.step(n); Bc.step(n)      # (in the same two calls as below: the first and last step of a call go through the general step kernel, whose sums are grouped differently)
    c0 = Bc.counters()
    assert c0["replans"] == 0 and c0["tiles_per_list"] == 5 and c0["nonbonded_kernel"] == 2 and c0["straggled"] == 0, c0
    ref = [(_state(g)) for g in calm]
    Bc.close()
    for g in calm:
        g.close()

    engs, B = build()
    B.step(n)
    x_before = engs[0].get_positions()
    engs[0].set_positions(_swap_waters_outward(s, x_before))
    B.step(n)
    c = B.counters()
    assert c["straggled"] == 1 and c["stragglers"] == 1 and c["replans"] == 0 and c["relayouts"] == 0, c
    assert c["nonbonded_kernel"] == 2 and c["tiles_per_list"] == 5, c                  # the batch keeps its layout ...
    assert engs[0].stats()["nonbonded_kernel"] == 3                                     # ... the straggler has its own
    assert engs[0].get_global("step") == 2 * n and np.isfinite(engs[0].get_global("protocol_work"))
    for g in engs:
        assert g.audit_lists()[1] == 0
    for g, (xr, vr, wr) in list(zip(engs, ref))[1:]:
        xg, vg, wg = _state(g)
        assert wg == wr, (wg, wr)
        assert np.array_equal(xg, xr), (np.abs(xg - xr).max(), int((np.abs(xg - xr).max(axis=1) > 0).sum()), np.nonzero(np.abs(xg - xr).max(axis=1) > 0)[0][:10])
        assert np.array_equal(vg, vr), np.abs(vg - vr).max()           # bit for bit: nobody else noticed
    # the next switch starts from a compact arrangement again (what a restored State brings): the straggler comes back
    B.reset_all()
    engs[0].set_positions(x_before)
    B.step(n)
    c2 = B.counters()
    assert c2["rejoined"] == 1 and c2["stragglers"] == 0 and c2["replans"] == 0, c2
    assert [g.stats()["nonbonded_kernel"] for g in engs] == [2] * R
    assert B.stats()["fallback_steps"] == 0      # (the member that came back took ONE step on launches of its own -- its lists were forced -- the others stayed in lock step)
    for g in engs:
        assert g.audit_lists()[1] == 0
    B.close()
    for g in engs:
        g.close()


def test_a_member_without_a_layout_does_not_cost_the_batched_energies(Engine, tune):
    """The head of a BLUES iteration restores every chain's State and asks for the energies of all of them (blues_batch_prefetch_energies).
    A member whose State arrives far from where its tiles were last laid out -- it was re-sorted in a spread arrangement during the
    previous switch -- has no layout at that point (resolve_xfer drops it), and the prefetch used to give up: every member then
    evaluated on demand, 1024 lone energy evaluations in an iteration that named no layout event.  It is laid out at the head of the
    prefetch now (counted in `relayouts`) and the OTHER members' energies come from the batched launches (none of them launches an
    evaluation of its own)."""
    from blues_amd.engine import NativeBatch
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    R, n = 6, 10
    tune(assume_batch=64)
    engs = []
    for r in range(R):
        g = Engine(s, _integ(100, seed=300 + r).to_data(precision=0, replica=r)); g.set_velocities(v * (1.0 + 0.02 * r)); engs.append(g)
    B = NativeBatch(engs)
    B.step(n)
    snaps = B.snapshot_all()                                         # the States the next iteration starts from (compact arrangement)
    engs[2].set_positions(_swap_waters_outward(s, engs[2].get_positions(), n_pairs=1, distance=1.3))
    B.step(n)                                                        # member 2 is laid out again for the spread arrangement (or leaves the batch's layout)
    B.reset_all()
    B.restore_all(snaps)                                             # ... and gets its compact State back: far from that layout
    c0, own0 = B.counters(), [g.stats()["own_energy_evaluations"] for g in engs]
    B.prefetch_energies(potential=True, kinetic=False)
    e_batched = [g.potential_energy() for g in engs]
    c1, own1 = B.counters(), [g.stats()["own_energy_evaluations"] for g in engs]
    assert c1["relayouts"] >= c0["relayouts"] + 1 or c1["rejoined"] > c0["rejoined"] or c1["stragglers"] > 0, (c0, c1)
    others = [r for r in range(R) if r != 2]
    assert [own1[r] for r in others] == [own0[r] for r in others], (own0, own1)     # served from the batched launches: nobody else evaluated alone
    ref = []
    for r, g in enumerate(engs):                                    # the same numbers from lone evaluations of the same states
        lone = Engine(s, _integ(100, seed=300 + r).to_data(precision=0, replica=r))
        lone.set_positions(g.get_positions())
        for name in ("lambda_sterics", "lambda_electrostatics"):     # (reset() leaves the context parameters where the last switch put them, as OpenMM does)
            lone.set_global(name, g.get_global(name))
        ref.append(lone.potential_energy())
        lone.close()
    B.close()
    for r in range(R):
        assert e_batched[r] == pytest.approx(ref[r], rel=2e-7, abs=1e-3), r          # (mixed precision: fp32 pair sums in another layout)
    for g in engs:
        g.close()


def test_a_batch_whose_shape_is_outgrown_by_many_moves_on_in_one_sweep(Engine, tune):
    """More members outgrow the shape than may straggle (one in 32; here: the second of six): the batch re-plans for everybody -- in
    the benchmark geometry that means fragment lists -- in ONE sweep over the members on the host's cores (round 5: up to three),
    counted and timed (blues_batch_get_counters), the lists complete."""
    from blues_amd.engine import NativeBatch
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    R, n = 6, 20
    tune(assume_batch=64)
    engs = []
    for r in range(R):
        g = Engine(s, _integ(100, seed=300 + r).to_data(precision=0, replica=r)); g.set_velocities(v * (1.0 + 0.02 * r)); engs.append(g)
    B = NativeBatch(engs)
    B.step(n)
    for r in (0, 1):
        engs[r].set_positions(_swap_waters_outward(s, engs[r].get_positions()))
    B.step(n)
    c = B.counters()
    assert c["replans"] == 1 and c["nonbonded_kernel"] == 3 and c["stragglers"] == 0, c
    assert R - 1 <= c["relayouts"] <= R and c["replan_seconds"] < 1.0, c              # one sweep: every member laid out once (the straggler had its layout)
    for g in engs:
        assert g.audit_lists()[1] == 0
        assert g.stats()["nonbonded_kernel"] == 3
    B.step(n)
    assert B.stats()["fallback_steps"] <= 2
    B.close()
    for g in engs:
        g.close()
