"""GPU suite: the BENCHMARKED configuration (S23k, 276 mobile atoms, mixed precision, batched, XCD-aware map) against the
committed oracle vectors tests/golden/s23k_oracle_vectors.npz (generator: tests/golden/make_s23k_vectors.py) --
BASELINE.json configs[0] to the letter: a 100-step switch with the rigid ligand rotation at step 50.

Bar (north_star): energies, forces and protocol work within 1e-5 relative of the fp64 CPU restatement.
  * double: pointwise over the whole switch.
  * mixed : energies / forces at 4 lambda pairs to 1e-5; the work trace to max|dw| <= 1e-5 max|w| teacher-forced every
            10 steps from the committed states (a liquid amplifies a 1e-7 force difference by ~e^(7.5/ps t): over the 0.4 ps
            of this switch two correct implementations drift apart pointwise, and the free-running trace is held to a
            looser, stated bound).
  * lone engine AND a replica batch of 8 (XCD map active, large-batch decomposition), every member against the vectors.

The 5000-step schedule (configs[4]'s purpose: accumulator accuracy over a long protocol) is teacher-forced against the live
oracle on the 975-atom box in mixed precision.
"""
import os

import numpy as np
import pytest

from blues_amd import integrators, systems

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "s23k_oracle_vectors.npz")


@pytest.fixture(scope="module")
def Engine():
    from blues_amd import build
    build.build_engine()
    from blues_amd.engine import NativeEngine
    return NativeEngine


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


@pytest.fixture(scope="module")
def s23k():
    return systems.s23k(mobile_atoms=275, frozen=True)


def _data(gold, precision, replica=0):
    return integrators.generateNCMCIntegrator(nstepsNC=int(gold["nsteps"]), dt=float(gold["dt"]), temperature=float(gold["temperature"]),
                                              seed=int(gold["seed"])).to_data(precision=precision, replica=replica)


@pytest.mark.parametrize("precision,tol", [(1, 1e-10), (0, 1e-5)])
def test_energies_and_mobile_forces_golden(Engine, gold, s23k, precision, tol):
    s, v = s23k
    mob = gold["mobile_atoms"]
    assert np.array_equal(mob, np.nonzero(s.mass > 0)[0])
    g = Engine(s, _data(gold, precision))
    for k, (ls, le) in enumerate(gold["lambdas"]):
        g.set_global("lambda_sterics", ls); g.set_global("lambda_electrostatics", le)
        t = g.energy_terms()
        assert abs(t.sum() - gold["energy_total"][k]) <= tol * abs(gold["energy_total"][k])
        for q in range(8):
            assert abs(t[q] - gold["energy_terms"][k][q]) <= tol * max(1.0, abs(gold["energy_terms"][k][q])), (k, q)
        f = g.get_forces()[mob]
        fo = gold["forces_mobile"][k]
        assert np.abs(f - fo).max() <= tol * np.abs(fo).max(), (k, np.abs(f - fo).max() / np.abs(fo).max())
        assert np.linalg.norm(f - fo) <= tol * np.linalg.norm(fo)
    g.close()


def _apply_move(x, s, gold):
    lig = np.asarray(s.alchemical_atoms)
    m = s.mass[lig]
    com = (x[lig] * m[:, None]).sum(0) / m.sum()
    xn = x.copy()
    xn[lig] = (x[lig] - com) @ gold["move_rotation"].T + com + gold["move_shift"]
    return xn


@pytest.mark.parametrize("precision", [1, 0])
@pytest.mark.parametrize("batched", [False, True])
def test_100_step_switch_with_rotation_golden(Engine, gold, s23k, precision, batched, tune):
    from blues_amd.engine import NativeBatch
    s, v = s23k
    wo = gold["work_trace"]
    scale = np.abs(wo).max()
    assert scale > 1.0
    R = 8 if batched else 1
    # (batched: eight congruent chains get the engine's own large-batch policy -- the decomposition bench.py runs; nothing pinned)

    def make():
        engs = [Engine(s, _data(gold, precision)) for _ in range(R)]
        if batched:
            B = NativeBatch(engs)
            return engs, B, (lambda n: B.step(n, trace=True)[1])
        return engs, None, (lambda n: [engs[0].run_switch(n, trace=True)])

    # ---- free-running: the whole switch from the initial state only
    engs, B, stepper = make()
    w = _run_switch_free(engs, stepper, s, v, gold)
    free_err = max(np.abs(wr - wo).max() for wr in w) / scale
    for wr in w[1:]:
        assert np.array_equal(wr, w[0])                 # identical chains of a batch stay bitwise identical
    xg = engs[0].get_positions()[gold["mobile_atoms"]]
    pos_err = np.abs(xg - gold["checkpoint_x"][-1]).max()
    if B is not None:
        assert B.stats()["fallback_steps"] <= 2         # the Move's setPositions is per member; everything else in lock step
        B.close()
    for g in engs:
        g.close()
    if precision == 1:
        assert free_err <= 1e-8 and pos_err < 1e-8, (free_err, pos_err)
    else:
        assert free_err <= 2e-4, free_err                # chaos-limited bound for 0.4 ps in mixed precision (stated, not 1e-5)
    # ---- teacher-forced every 10 steps: north_star's 1e-5 on the work, mixed precision included
    engs, B, stepper = make()
    w = _run_switch_teacher(engs, stepper, s, v, gold)
    err = max(np.abs(wr - wo).max() for wr in w) / scale
    if B is not None and precision == 0:
        # a batch of 8 in mixed precision is checked against the golden vectors THROUGH the dense alchemical kernel and the per-atom-list
        # nonbonded kernel -- what bench.py runs -- not through the forms a lone chain uses
        st = engs[0].stats()
        assert st["alchemical_kernel"] == 1 and st["nonbonded_kernel"] == 2, st
    if B is not None:
        B.close()
    for g in engs:
        g.close()
    assert err <= (1e-9 if precision == 1 else 1e-5), (err, free_err)
    print("S23k configs[0] switch: precision=%d batched=%s free-running %.2e teacher-forced %.2e (of max|w| = %.3f kJ/mol)" % (precision, batched, free_err, err, scale))


def test_bench_configuration_golden(Engine, gold, s23k):
    """The launch configuration bench.py runs, with NOTHING pinned: 256 congruent S23k chains in one replica batch under the engine's
    own policy (default BluesTuning: per-atom lists pruned in passing, default margins and alchemical block shape, side-stream fork),
    mixed precision, BASELINE.json configs[0] -- the 100-step switch with the rigid rotation at step 50 -- teacher-forced every 10
    steps from the committed oracle states: protocol work within 1e-5 of max|w| for every member, and members that were given
    identical inputs stay bitwise identical (they share launches but never data).  bench.py's default is two such batches of 1024 chains taking
    turns on the device (test_two_batches_taking_turns_meet_the_golden_vectors below): the same code path and decomposition (the layout
    changes at n_itiles * R > 32); 256 keeps this test's set-up time down."""
    from blues_amd import tuning
    from blues_amd.engine import NativeBatch
    assert tuning.as_dict() == {k: getattr(tuning.defaults(), k) for k in tuning.FIELDS}       # no override is active
    s, v = s23k
    R = 256
    wo = gold["work_trace"]; scale = np.abs(wo).max()
    engs = [Engine(s, _data(gold, 0)) for _ in range(R)]
    B = NativeBatch(engs)
    w = _run_switch_teacher(engs, lambda n: B.step(n, trace=True)[1], s, v, gold)
    st, bst = engs[0].stats(), B.stats()
    assert st["nonbonded_kernel"] == 2 and st["pruned_lists"] == 1 and st["atom_prunes"] >= 100, st     # the benchmarked kernel, pruning as it goes
    assert st["alchemical_kernel"] == 1, st    # ... and the DENSE alchemical kernel (what large batches run): a flipped default would move this test to the other form
    assert bst["fallback_steps"] <= 2 * (int(gold["nsteps"]) // int(gold["checkpoint_every"])), bst       # only the re-synchronisations are per member
    errs = [np.abs(w[r] - wo).max() / scale for r in (0, 1, 7, 8, 31, 64, 100, 127, 128, 200, 254, 255)]
    assert max(errs) <= 1e-5, errs
    for r in range(1, R):
        assert np.array_equal(w[r], w[0]), r
    x0 = engs[0].get_positions()
    for r in (1, 128, 255):
        assert np.array_equal(engs[r].get_positions(), x0)
    B.close()
    for g in engs:
        g.close()
    print("bench configuration (R = %d, default policy) teacher-forced work error %.2e of max|w| = %.3f kJ/mol; %d per-atom prunes, %d rebuilds per chain"
          % (R, max(errs), scale, st["atom_prunes"], st["list_generation"]))


def test_benchmarked_system_through_fragment_lists_golden(Engine, gold, s23k, tune):
    """What a real run's NCMC engines become once their mobile atoms have scattered over MD legs (the engine re-lays
    itself out on its own, tests/test_gpu_frag.py::test_scattered_mobile_atoms_fall_back_to_fragment_lists): the benchmarked System
    -- 276 mobile atoms, everything else frozen -- through the fragment lists (i-side: the 97 fragments that hold a mobile atom; frozen
    j-fragments with half the margins), mixed precision.  Energies, term sums and mobile-atom forces at the four committed lambda
    pairs within 1e-5, and configs[0]'s 100-step switch teacher-forced in a batch of 8: work within 1e-5 of max|w|, lists audited."""
    from blues_amd.engine import NativeBatch
    s, v = s23k
    mob = gold["mobile_atoms"]
    tune(k1_mode=3)
    g = Engine(s, _data(gold, 0))
    assert g.stats()["nonbonded_kernel"] == 3, g.stats()
    for k, (ls, le) in enumerate(gold["lambdas"]):
        g.set_global("lambda_sterics", ls); g.set_global("lambda_electrostatics", le)
        t = g.energy_terms()
        assert abs(t.sum() - gold["energy_total"][k]) <= 1e-5 * abs(gold["energy_total"][k])
        for q in range(8):
            assert abs(t[q] - gold["energy_terms"][k][q]) <= 1e-5 * max(1.0, abs(gold["energy_terms"][k][q])), (k, q)
        f = g.get_forces()[mob]
        fo = gold["forces_mobile"][k]
        assert np.abs(f - fo).max() <= 1e-5 * np.abs(fo).max(), (k, np.abs(f - fo).max() / np.abs(fo).max())
    assert g.audit_lists()[1] == 0
    g.close()
    R = 8
    tune(k1_mode=3, assume_batch=R)
    wo = gold["work_trace"]; scale = np.abs(wo).max()
    engs = [Engine(s, _data(gold, 0)) for _ in range(R)]
    B = NativeBatch(engs)
    w = _run_switch_teacher(engs, lambda n: B.step(n, trace=True)[1], s, v, gold)
    err = max(np.abs(wr - wo).max() for wr in w) / scale
    assert engs[0].stats()["nonbonded_kernel"] == 3
    for e in (engs[0], engs[-1]):
        assert e.audit_lists()[1] == 0
    for wr in w[1:]:
        assert np.array_equal(wr, w[0])
    B.close()
    for e in engs:
        e.close()
    assert err <= 1e-5, err
    print("benchmarked System through fragment lists: teacher-forced work error %.2e of max|w| = %.3f kJ/mol" % (err, scale))


def test_two_batches_taking_turns_meet_the_golden_vectors(Engine, gold, s23k):
    """The ARRANGEMENT bench.py runs by default: two replica batches on one GPU, each driven from its own host thread, their stepping
    calls taking turns on the device under one lock (simulation.BatchedBLUESSimulation(device_turn=...), engine.NativeBatch.device_turn).
    Both batches run the teacher-forced golden switch at the same time: every member within 1e-5 of max|w|, members bitwise equal."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from blues_amd import tuning
    from blues_amd.engine import NativeBatch
    assert tuning.as_dict() == {k: getattr(tuning.defaults(), k) for k in tuning.FIELDS}
    s, v = s23k
    R = 128
    wo = gold["work_trace"]; scale = np.abs(wo).max()
    turn = threading.Lock()
    groups = []
    for g in range(2):
        engs = [Engine(s, _data(gold, 0)) for _ in range(R)]
        B = NativeBatch(engs); B.device_turn = turn
        groups.append((engs, B))

    def run(g):
        engs, B = groups[g]
        return _run_switch_teacher(engs, lambda n: B.step(n, trace=True)[1], s, v, gold)
    with ThreadPoolExecutor(max_workers=2) as pool:
        ws = list(pool.map(run, range(2)))
    for (engs, B), w in zip(groups, ws):
        st = engs[0].stats()
        assert st["nonbonded_kernel"] == 2 and st["alchemical_kernel"] == 1 and st["pruned_lists"] == 1, st
        errs = [np.abs(w[r] - wo).max() / scale for r in (0, 1, 63, 64, 127)]
        assert max(errs) <= 1e-5, errs
        for r in range(1, R):
            assert np.array_equal(w[r], w[0]), r
    assert np.array_equal(ws[0][0], ws[1][0])       # the two batches ran the same chains: the turns change no bit
    for engs, B in groups:
        B.close()
        for g in engs:
            g.close()


def _run_switch_free(engines, stepper, s, v, gold):
    n, move_step = int(gold["nsteps"]), int(gold["move_step"])
    for g in engines:
        g.set_velocities(v)
    a = stepper(move_step)
    for g in engines:
        g.set_positions(_apply_move(g.get_positions(), s, gold))
    b = stepper(n - move_step)
    return [np.concatenate([np.asarray(a[r]), np.asarray(b[r])]) for r in range(len(engines))]


def _run_switch_teacher(engines, stepper, s, v, gold):
    """Segments of `checkpoint_every` steps, each started from the committed (x, v) of the oracle; compared through the
    work INCREMENTS of every segment, accumulated on the oracle's level.  A re-synchronisation is an instantaneous edit:
    the engine books U(x_gold) - U(x_gpu) for it in the first step of the segment (perturbed_pe - unperturbed_pe,
    reference blues/integrators.py:184-191), which is taken out again.  At the move step the re-sync and the Move are two
    setPositions calls; the integrator sees their sum, so the Move's own work U(x_moved) - U(x_gold) -- the part the oracle
    booked as well -- is evaluated explicitly and kept."""
    n, every, move_step = int(gold["nsteps"]), int(gold["checkpoint_every"]), int(gold["move_step"])
    mob = gold["mobile_atoms"]
    wo = gold["work_trace"]
    for g in engines:
        g.set_velocities(v)
    out = [np.zeros(n) for _ in engines]
    raw_prev = [0.0 for _ in engines]
    for seg, start in enumerate(range(0, n, every)):
        move_work = [0.0 for _ in engines]
        for r, g in enumerate(engines):
            if not start:
                continue
            x = g.get_positions(); vv = g.get_velocities()
            x[mob] = gold["checkpoint_x"][seg]; vv[mob] = gold["checkpoint_v"][seg]
            g.set_positions(x); g.set_velocities(vv)
            if start == move_step:
                e0 = g.potential_energy()
                xm = _apply_move(x, s, gold)
                assert np.abs(xm[mob] - gold["x_after_move_mobile"]).max() < 1e-12
                g.set_positions(xm)
                move_work[r] = g.potential_energy() - e0
        tr = stepper(every)
        for r, g in enumerate(engines):
            t = np.asarray(tr[r], dtype=np.float64)
            d = np.diff(np.concatenate([[raw_prev[r]], t]))   # the engine's accumulator runs on across segments
            if start:
                booked = g.get_global("perturbed_pe") - g.get_global("unperturbed_pe")   # re-sync (+ Move) as the integrator saw it
                d[0] -= booked - move_work[r]
            out[r][start:start + every] = (wo[start - 1] if start else 0.0) + np.cumsum(d)
            raw_prev[r] = t[-1]
    return out


def test_5000_step_schedule_accumulator_mixed_precision(Engine, oracle_mod):
    """configs[4]'s purpose: accuracy of the running protocol-work accumulator over a 5000-step schedule (10,000 lambda
    increments of 1e-4) in the benchmarked MIXED precision.  975-atom box, live oracle, teacher-forced every 10 steps (see
    the module docstring); the accumulated work must stay within 1e-5 of the work scale over the whole protocol, and the
    sum of the per-segment discrepancies must not drift (no systematic bias in the accumulation)."""
    s, v = systems.toluene_box()
    n, seg, tol = 5000, 10, 1e-5
    data = integrators.generateNCMCIntegrator(nstepsNC=n, dt=0.004, temperature=300.0, seed=77).to_data(precision=0)
    g, o = Engine(s, data), oracle_mod.Oracle(s, data)
    g.set_velocities(v); o.set_velocities(v)
    scale, worst, drift = 0.0, 0.0, 0.0
    wg_prev = wo_prev = 0.0
    for start in range(0, n, seg):
        if start:
            g.set_positions(o.get_positions()); g.set_velocities(o.get_velocities())
        wg = g.run_switch(seg, trace=True)
        wo = np.empty(seg)
        for k in range(seg):
            o.step(1); wo[k] = o.get_global("protocol_work")
        dg = np.diff(np.concatenate([[wg_prev], wg])); do = np.diff(np.concatenate([[wo_prev], wo]))
        if start:
            dg[0] -= g.get_global("perturbed_pe") - g.get_global("unperturbed_pe")
        worst = max(worst, np.abs(np.cumsum(dg) - np.cumsum(do)).max())
        drift += dg.sum() - do.sum()
        scale = max(scale, np.abs(wo).max())
        wg_prev, wo_prev = wg[-1], wo[-1]
    assert scale > 10.0
    assert worst <= tol * scale, (worst, scale)
    assert abs(drift) <= 10 * tol * scale, (drift, scale)     # 500 segments: the per-segment errors do not add up coherently
    assert g.get_global("lambda") == pytest.approx(1.0) and g.get_global("step") == n
    g.close()
