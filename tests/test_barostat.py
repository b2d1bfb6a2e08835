"""MonteCarloBarostat of the MD leg (reference blues/simulation.py:603-626), host logic on the oracle-backed test double (CPU);
the same sequence on the HIP engine is in tests/test_gpu_parity.py::test_barostat_on_gpu_matches_the_oracle."""
import copy

import numpy as np
import pytest

from blues_amd import barostat, integrators, systems, unit


def test_molecules_and_rigid_scaling(tol_box):
    s, _ = tol_box
    mol = barostat.molecules_of(s)
    assert mol.max() + 1 == 321 and np.all(mol[:15] == mol[0]) and np.all(mol[15:18] == mol[15]) and mol[18] != mol[15]   # toluene + 320 waters
    b = barostat.MonteCarloBarostat(1.0, 300.0, 25, seed=1)
    x = s.positions
    y = b.scaled_positions(x, mol, 1.01)
    for m in (0, 7, 320):           # intramolecular geometry untouched, centres scaled
        a = np.nonzero(mol == m)[0]
        assert np.allclose(y[a] - y[a[0]], x[a] - x[a[0]], atol=1e-14) and np.allclose(y[a].mean(0), 1.01 * x[a].mean(0), atol=1e-12)
    assert barostat.BAR_NM3 == pytest.approx(6.02214076e23 * 1e5 * 1e-27 * 1e-3, rel=1e-12)


def test_volume_moves_through_the_simulation_mirror(oracle_backed_context, tol_box):
    """Simulation.step makes an attempt before every frequency-th step (OpenMM's updateContextState counts at the top of each step and
    acts when the count reaches `frequency`); frequency 0 disables the barostat; the Metropolis weight is dU + P dV - N kT ln(V'/V); a rejected move
    restores box and coordinates exactly; the step size adapts."""
    ctx = oracle_backed_context
    s, v = tol_box
    md = copy.copy(s); md.alchemical_atoms = np.zeros(0, np.int32)
    md = systems.add_barostat(md, 300.0, 1.0, 5)
    sim = ctx.Simulation(None, md, integrators.LangevinIntegrator(300.0, 1.0, 0.002, seed=3))
    sim.context.setVelocities(unit.Quantity(v, "nanometer/picosecond"))
    eng = sim.context._engine
    assert sim.barostat is not None and sim.barostat.frequency == 5
    vol0 = np.prod(np.diag(eng.get_box()))
    seen = []
    for blk in range(12):
        x_before, box_before = eng.get_positions(), eng.get_box().copy()
        sim.step(5)
        if sim.barostat.last is not None and sim.barostat.total_attempted > len(seen):
            seen.append(dict(sim.barostat.last))
    assert sim.barostat.total_attempted == 12           # before steps 5, 10, ..., 60 (after frequency - 1 completed steps, then every frequency)
    assert sim.currentStep == 60
    acc = [r for r in seen if r["accepted"]]
    assert 0 < len(acc) <= len(seen)
    for r in seen:
        n = 321
        # the recorded weight is what the rule says it is
        assert r["w"] == pytest.approx(r["dU"] + sim.barostat.pressure * r["dV"] - n * sim.barostat.kT * np.log((r["volume"] if r["accepted"] else r["volume"] + r["dV"]) / (r["volume"] - r["dV"] if r["accepted"] else r["volume"])), rel=1e-9, abs=1e-9)
    vol1 = np.prod(np.diag(eng.get_box()))
    assert vol1 == pytest.approx(seen[-1]["volume"], rel=1e-12) and abs(vol1 / vol0 - 1) < 0.2
    # frequency 0 = no barostat: stepping neither hangs nor attempts
    sim.barostat.frequency = 0
    sim.step(7)
    assert sim.currentStep == 67 and sim.barostat.total_attempted == 12
    sim.barostat.frequency = 5
    # a rejected attempt leaves the state bit for bit where it was
    b = sim.barostat
    b.pressure = 1e9          # an absurd pressure rejects every expansion
    b.rng = np.random.RandomState(0)
    while True:
        x0, box0 = eng.get_positions(), eng.get_box().copy()
        if not b.attempt(eng, md):
            break
    assert np.array_equal(eng.get_positions(), x0) and np.array_equal(eng.get_box(), box0)
