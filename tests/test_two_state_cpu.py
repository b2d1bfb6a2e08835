"""CPU dress rehearsal of tests/test_gpu_statistics.py::test_two_state_populations_through_the_batched_driver: the same two-state
system (tests/two_state.py: populations known by quadrature of its formula, no oracle involved) through the same driver on
oracle-backed test doubles.  For the ORACLE this is a known answer it did not produce itself: the step program, the work of the
instantaneous move, the Metropolis test, the velocity re-draw and the MD leg have to be right for the populations to come out."""
import numpy as np
import pytest

import two_state as ts
from conftest import OracleBackedEngine
from test_batched_driver_cpu import OracleBackedBatch


def test_the_quadrature_itself():
    p1 = ts.exact_populations(300.0, h=0.002)
    p2 = ts.exact_populations(300.0, h=0.004)
    assert abs(p1[0] + p1[1] - 1.0) < 1e-12 and abs(p1[0] - p2[0]) < 1e-6      # converged in the grid spacing
    assert 0.30 < p1[0] < 0.37                                                    # well away from 1/2: an acceptance rule that ignores the work gives 1/2
    # harmonic estimate of the ratio: exp(dE / kT) with the wells' depths, times the ratio of the radial widths sqrt(e_P / e_Q)
    kT = ts.KB * 300.0
    est = np.exp((18.0 - 16.0) / kT) * np.sqrt(16.0 / 18.0)
    assert abs(p1[1] / p1[0] - est) < 0.15 * est


def test_two_state_populations_on_the_oracle(monkeypatch):
    from blues_amd import context, engine, simulation
    monkeypatch.setattr(context, "NativeEngine", OracleBackedEngine)
    monkeypatch.setattr(engine, "NativeBatch", OracleBackedBatch)
    R, n_iter, burn = 96, 14, 5
    chains = [ts.build_chain(context, r, 100, 50, 300.0, 0.002, precision="double") for r in range(R)]
    B = simulation.BatchedBLUESSimulation(chains)
    in_q = []
    for it in range(n_iter):
        B.run(nIter=1, nstepsNC=100, moveStep=50, nstepsMD=50)
        in_q.append(ts.basin_of(np.array([c._md_sim.context._engine.get_positions()[0] for c in chains])))
    p_exact = ts.exact_populations(300.0)[1]
    pooled = np.array(in_q[burn:], dtype=float)
    # chains are independent; within a chain successive iterations are anticorrelated (most proposals are accepted): the error of
    # the pooled mean is taken from the spread of the per-chain time averages
    per_chain = pooled.mean(axis=0)
    se = per_chain.std(ddof=1) / np.sqrt(R)
    assert se < 0.04
    assert abs(per_chain.mean() - p_exact) < 4.0 * se, (per_chain.mean(), p_exact, se)
    assert abs(per_chain.mean() - 0.5) > 4.0 * se            # ... and the test can tell the answer from "every proposal accepted"
    acc = sum(c.accept for c in chains) / float(R * n_iter)
    assert 0.3 < acc < 0.95
