"""Oracle-INDEPENDENT evidence on the HIP engine: statistical identities and a known answer that follow from the published
formulas alone (SURVEY.md section 7 names Jarzynski / detailed-balance checks as the substitute for golden vectors the reference does
not hold; /root/reference/blues/tests/test_ethylene.py:107-163 is the reference's own test of this kind, on the real engine).
Nothing here calls the oracle.

1. Jarzynski's equality for a SYMMETRIC protocol: chains that start in equilibrium at lambda = 0 and are switched out and back
   (lambda = 0 and lambda = 1 are the same Hamiltonian) must satisfy <exp(-W / kT)> = 1, whatever the dynamics in between, if and only
   if W is the sum of the energy changes at fixed coordinates every time the parameters change -- the definition the integrator's H
   step implements (/root/reference/blues/integrators.py:211-231).  A term left out, a lambda table read one entry off, an energy
   difference formed at moved coordinates: each shows up as a mean different from 1.
2. A two-state system with populations known by quadrature, through BatchedBLUESSimulation with MD legs (tests/two_state.py)."""
import numpy as np
import pytest

from conftest import gpu_available

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not gpu_available(), reason="needs a GPU")]

from blues_amd import integrators, systems  # noqa: E402

KB = 0.0083144626


def _bootstrap_se(values, fn=np.mean, n=400, seed=0):
    rs = np.random.RandomState(seed)
    values = np.asarray(values)
    return float(np.std([fn(values[rs.randint(0, len(values), len(values))]) for _ in range(n)], ddof=1))


def test_jarzynski_equality_for_a_symmetric_protocol():
    """1024 chains of the 975-atom toluene box (everything mobile, mixed precision: what bench.py runs), one engine per chain whose
    protocol is flat for the first N_EQ steps -- plain Langevin dynamics at full interaction: every chain forgets the common start under
    its own noise -- and then takes lambda_sterics 1 -> 0.85 -> 1 and lambda_electrostatics 1 -> 0.7 -> 1 along a half sine over N_SW
    steps (gentle: the work is of the order of kT, so the exponential average converges).  dt = 1 fs, friction 5 / ps."""
    from blues_amd import build
    build.build_engine()
    from blues_amd.engine import NativeBatch, NativeEngine
    from blues_amd.replicas import build_in_parallel, replica_seed
    s, v = systems.toluene_box()
    R, N_EQ, N_SW, T = 1024, 3000, 600, 300.0
    kT = KB * T
    n_total = N_EQ + N_SW
    l0 = N_EQ / float(n_total)
    prog = "max(0, (lambda-%.17g)/(1-%.17g))" % (l0, l0)
    functions = {"lambda_sterics": "1 - 0.15*sin(3.141592653589793*%s)" % prog, "lambda_electrostatics": "1 - 0.3*sin(3.141592653589793*%s)" % prog}

    def make(r):
        integ = integrators.AlchemicalExternalLangevinIntegrator(functions, splitting="H V R O R V H", temperature=T, collision_rate=5.0, timestep=0.001,
                                                                 nsteps_neq=n_total, seed=replica_seed(99, r))
        g = NativeEngine(s, integ.to_data(precision=0, replica=r))
        g.set_velocities_to_temperature(T, 1000 + r)
        return g
    engs = build_in_parallel(make, R)
    B = NativeBatch(engs)
    B.step(N_EQ)
    w_eq = np.array([g.get_global("protocol_work") for g in engs])
    assert np.abs(w_eq).max() < 1e-9                     # a flat protocol does no work: E(x; same parameters) - E(x; same parameters), term by term
    _, trace = B.step(N_SW, trace=True)
    assert B.stats()["fallback_steps"] == 0
    W = trace[:, -1] / kT                                # whole symmetric protocol, in kT
    W_half = trace[:, N_SW // 2 - 1] / kT                # the outward half alone (a protocol with dF > 0)
    B.close()
    for g in engs:
        g.close()
    assert np.isfinite(W).all()
    x = np.exp(-W)
    m, se = x.mean(), _bootstrap_se(x)
    assert 0.05 < W.std() < 2.0, W.std()                 # gentle, but not a protocol that does nothing
    assert W.mean() > 0.0                                # second law (Jensen, given the equality)
    assert se < 0.05, se
    assert abs(m - 1.0) < 4.0 * se, (m, se, W.mean(), W.std())
    # what the equality rules out, with the same estimator: a bias of a quarter kT in every W; the work taken with the other sign; and
    # the half protocol, whose average is exp(-dF) with dF of several kT (the ligand's dispersion attraction partly switched off) --
    # the whole protocol's value of 1 is the cancellation of two such halves
    assert abs((x * np.exp(-0.25)).mean() - 1.0) > 4.0 * se
    xr = np.exp(W)
    assert xr.mean() - 1.0 > 4.0 * _bootstrap_se(xr)
    xh = np.exp(-W_half)
    assert abs(xh.mean() - 1.0) > 10.0 * _bootstrap_se(xh) and xh.mean() < 0.5, xh.mean()
    # Crooks for a symmetric protocol, P(W) / P(-W) = exp(W): <1[W > 0] exp(-W)> = P(W < 0) -- a statement about the shape of the
    # distribution, not only its exponential mean
    lhs = (W > 0) * np.exp(-W)
    rhs = (W < 0).astype(float)
    d = lhs - rhs
    assert abs(d.mean()) < 4.0 * _bootstrap_se(d) + 1e-3, (lhs.mean(), rhs.mean())


def test_two_state_populations_through_the_batched_driver():
    """2048 chains of tests/two_state.py through BatchedBLUESSimulation.run with MD legs -- the batched plugin boundary, the NCMC engine's
    alchemical kernel, the MD engine's nonbonded kernel, frozen atoms, the restraint, the Metropolis test -- against populations known
    by quadrature: p(Q side) = 0.6644 at 300 K.  An acceptance rule that ignores the work gives 0.5."""
    import two_state as ts
    from blues_amd import build, context, simulation
    build.build_engine()
    from blues_amd.replicas import build_in_parallel
    R, n_iter, burn = 2048, 16, 6
    chains = build_in_parallel(lambda r: ts.build_chain(context, r, 100, 50, 300.0, 0.002, precision="mixed"), R)
    B = simulation.BatchedBLUESSimulation(chains)
    assert B._batchable() and B._move_batchable()
    in_q = []
    for it in range(n_iter):
        B.run(nIter=1, nstepsNC=100, moveStep=50, nstepsMD=50)
        x = B._md_batch.read_atoms_all([0])[:, 0, :]
        in_q.append(ts.basin_of(x))
    p_exact = ts.exact_populations(300.0)[1]
    per_chain = np.array(in_q[burn:], dtype=float).mean(axis=0)
    se = per_chain.std(ddof=1) / np.sqrt(R)
    acc = sum(c.accept for c in chains) / float(R * n_iter)
    B.close()
    assert se < 0.01
    assert abs(per_chain.mean() - p_exact) < 4.0 * se, (per_chain.mean(), p_exact, se, acc)
    assert abs(per_chain.mean() - 0.5) > 10.0 * se
    assert 0.3 < acc < 0.95, acc
    last = np.asarray(in_q[-1], dtype=float)
    assert abs(last.mean() - p_exact) < 4.0 * np.sqrt(p_exact * (1 - p_exact) / R)
