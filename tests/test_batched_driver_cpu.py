"""CPU suite: the lock-step driver (blues_amd.simulation.BatchedBLUESSimulation) on oracle-backed test doubles -- hook order,
per-chain error isolation, masked-out chains, reporter intervals -- against the same chains run one after the other through
BLUESSimulation.  The native batch itself is covered on the GPU (tests/test_gpu_batch.py)."""
import numpy as np
import pytest

from conftest import OracleBackedEngine


class OracleBackedBatch:
    """TEST DOUBLE with engine.NativeBatch's interface: steps each member's oracle in turn."""

    def __init__(self, engines):
        self.engines = list(engines)
        self.calls = []

    def __len__(self):
        return len(self.engines)

    def step(self, n=1, trace=False, raise_errors=True, active=None):
        self.calls.append((int(n), None if active is None else tuple(bool(a) for a in active)))
        errors = []
        for r, e in enumerate(self.engines):
            if active is not None and not active[r]:
                errors.append(None); continue
            try:
                if getattr(e, "fail_next", False):
                    e.fail_next = False
                    raise RuntimeError("Particle coordinate is nan")
                e.step(n); errors.append(None)
            except Exception as ex:  # noqa: BLE001 - mirrors the per-member status of the native batch
                errors.append(ex)
        if raise_errors:
            for ex in errors:
                if ex is not None:
                    raise ex
        return errors, None

    def prefetch_energies(self, potential=True, kinetic=True, active=None, at_lambda_one=False):
        pass

    def close(self):
        pass


@pytest.fixture()
def doubles(monkeypatch):
    from blues_amd import context, engine
    monkeypatch.setattr(context, "NativeEngine", OracleBackedEngine)
    monkeypatch.setattr(engine, "NativeBatch", OracleBackedBatch)
    return context


def _chain(context, s, v, r, nsteps, move_cls=None, nIter=1):
    from blues_amd import integrators, moves, simulation, unit
    integ = integrators.generateNCMCIntegrator(nstepsNC=nsteps, dt=0.002, temperature=300.0, seed=40 + r)
    sim = context.Simulation(None, s, integ, replica=r)
    sim.context.setVelocities(unit.Quantity(v * (1.0 + 0.03 * r), "nanometer/picosecond"))
    lig = np.arange(15)
    mv = (move_cls or moves.RandomLigandRotationMove)(lig, s.mass[lig], random_state=90 + r)
    return simulation.BLUESSimulation(simulation.SimulationSet(sim), {"nstepsNC": nsteps, "moveStep": nsteps // 2, "nIter": nIter}, moves.MoveEngine(mv))


def test_lockstep_driver_equals_separate_chains(doubles, tol_box):
    from blues_amd import simulation
    s, v = tol_box
    R, nsteps = 3, 6
    sep = [_chain(doubles, s, v, r, nsteps) for r in range(R)]
    for r, c in enumerate(sep):
        c._syncStatesMDtoNCMC(); c._stepNCMC(nsteps, nsteps // 2)
        np.random.seed(100 + r)
        c._acceptRejectMove()
    bat = [_chain(doubles, s, v, r, nsteps) for r in range(R)]
    B = simulation.BatchedBLUESSimulation(bat)
    for c in bat:
        c._syncStatesMDtoNCMC()
    B._stepNCMC(nsteps, nsteps // 2)
    assert [n for n, _ in B._ncmc_batch.calls] == [nsteps // 2, nsteps // 2]     # two shared step calls: before and after the move
    for r, c in enumerate(bat):
        np.random.seed(100 + r)
        c._acceptRejectMove()
        assert c.last == sep[r].last
        assert np.array_equal(c.stateTable["ncmc"]["state1"]["positions"]._value, sep[r].stateTable["ncmc"]["state1"]["positions"]._value)


def test_a_failing_chain_is_abandoned_alone(doubles, tol_box):
    """Chain 1's integrator raises in the steps after its move (what an OpenMMException "Particle coordinate is nan" is in the
    reference): the reference policy (log, move._error, abandon the switch, simulation.py:1088-1094) applies to that chain
    only; the others finish as if it were not there."""
    from blues_amd import moves, simulation

    class Exploding(moves.RandomLigandRotationMove):
        errors = 0
        def move(self, context):
            context._engine.fail_next = True      # the next integrator step of THIS chain raises
            return context
        def _error(self, context):
            Exploding.errors += 1
            return context

    s, v = tol_box
    nsteps = 6
    ref = _chain(doubles, s, v, 0, nsteps)
    np.random.seed(5); ref._syncStatesMDtoNCMC(); ref._stepNCMC(nsteps, nsteps // 2)
    bat = [_chain(doubles, s, v, 0, nsteps), _chain(doubles, s, v, 1, nsteps, move_cls=Exploding), _chain(doubles, s, v, 2, nsteps)]
    B = simulation.BatchedBLUESSimulation(bat)
    for c in bat:
        c._syncStatesMDtoNCMC()
    B._stepNCMC(nsteps, nsteps // 2)
    assert Exploding.errors == 1
    assert B._ncmc_batch.calls == [(nsteps // 2, (True, True, True)), (nsteps // 2, (True, True, True))]
    assert bat[1]._ncmc_sim.context._integrator.getGlobalVariableByName("step") == nsteps // 2      # never advanced past the failed hook
    assert np.array_equal(bat[0].stateTable["ncmc"]["state1"]["positions"]._value, ref.stateTable["ncmc"]["state1"]["positions"]._value)
    assert "state1" in bat[1].stateTable["ncmc"] and bat[1].stateTable["ncmc"]["state1"]      # the abandoned chain still records its end state


def test_reporter_intervals_split_the_shared_steps(doubles, tol_box):
    """A reporter on one chain's NCMC simulation is due every 2 steps: the shared step calls are cut there for everybody, the
    reporter sees its states, results do not change."""
    from blues_amd import simulation
    s, v = tol_box
    nsteps = 8

    class Every2:
        def __init__(self): self.seen = []
        def describeNextReport(self, sim): return (2 - sim.currentStep % 2, False, False, False, True)
        def report(self, sim, state): self.seen.append((sim.currentStep, state.getPotentialEnergy()._value))

    plain = [_chain(doubles, s, v, r, nsteps) for r in range(2)]
    Bp = simulation.BatchedBLUESSimulation(plain)
    for c in plain:
        c._syncStatesMDtoNCMC()
    Bp._stepNCMC(nsteps, nsteps // 2)
    rep = [_chain(doubles, s, v, r, nsteps) for r in range(2)]
    reporter = Every2()
    rep[1]._ncmc_sim.reporters.append(reporter)
    Br = simulation.BatchedBLUESSimulation(rep)
    for c in rep:
        c._syncStatesMDtoNCMC()
    Br._stepNCMC(nsteps, nsteps // 2)
    assert [n for n, _ in Br._ncmc_batch.calls] == [2, 2, 2, 2]
    assert [st for st, _ in reporter.seen] == [2, 4, 6, 8]
    for a, b in zip(plain, rep):
        assert np.array_equal(a.stateTable["ncmc"]["state1"]["positions"]._value, b.stateTable["ncmc"]["state1"]["positions"]._value)


def test_a_retired_chain_sits_out_the_chain_by_chain_path(doubles, tol_box):
    """isolate_failures on the path that is NOT batched (here: a test double without the batched plugin boundary; in production: reporters
    on the NCMC leg, batched_boundary=False): a retired chain is skipped by the sync, the switch, the Metropolis step and the reset, and
    the others -- every chain on a private RandomState, although rng=None was asked for -- come out exactly as in a run in which nobody
    was retired (the numbers a dead chain does not draw shift nobody's stream)."""
    from blues_amd import simulation
    s, v = tol_box
    R, nsteps = 3, 4

    def run(retire):
        np.random.seed(7)
        chains = [_chain(doubles, s, v, r, nsteps, nIter=2) for r in range(R)]
        B = simulation.BatchedBLUESSimulation(chains, isolate_failures=True)
        assert not B._batchable()
        assert all(c._rng is not np.random for c in chains)
        if retire is not None:
            B._retire(retire, RuntimeError("blown up"), "a test")
        B.run(nIter=2, nstepsNC=nsteps, moveStep=nsteps // 2)
        return B, chains

    B0, all_alive = run(None)
    B1, one_dead = run(1)
    assert sorted(B1.dead) == [1] and one_dead[1].last.get("failed")
    assert one_dead[1]._ncmc_sim.context._integrator.getGlobalVariableByName("step") == 0      # never stepped, never reset, never read
    assert (1, ) not in [tuple(i for i, a in enumerate(act) if a) for _, act in B1._ncmc_batch.calls if act is not None]
    for r in (0, 2):
        assert one_dead[r].last == all_alive[r].last
        assert one_dead[r].accept == all_alive[r].accept
        assert np.array_equal(one_dead[r]._ncmc_sim.context._engine.get_positions(), all_alive[r]._ncmc_sim.context._engine.get_positions())
