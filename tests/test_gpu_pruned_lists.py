"""Dual neighbour lists of the benchmark's nonbonded kernel (kernels_nb.h: nonbonded_atom_body) on a real MI355X.

The kernel walks PRUNED per-atom lists (cutoff + a small inner margin) and re-derives them from the full Verlet lists -- atom by atom,
whenever that atom has moved the inner margin.  What must hold, whatever the margins: at every step the pruned lists contain every pair
inside the cutoff.  Checked three ways on the benchmark system (S23k, 276 mobile atoms, mixed precision, the decomposition of
a large batch): (i) mid-life forces of a chain that has been pruning for a while against a fresh evaluation of the same
coordinates by an engine without pruned lists and against the fp64 oracle; (ii) the trajectories with and without pruned
lists stay together to summation-order rounding; (iii) the bookkeeping (prune passes happen, far more often than rebuilds).
Reference behaviour being reproduced: the NonbondedForce evaluation behind CustomIntegrator's `f`
(reference blues/integrators.py:159-231)."""
import numpy as np
import pytest

from blues_amd import build, integrators, systems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Engine():
    build.build_engine()
    from blues_amd.engine import NativeEngine
    return NativeEngine


def _integ(n, seed=7):
    return integrators.generateNCMCIntegrator(nstepsNC=n, dt=0.004, temperature=300.0, seed=seed)


@pytest.mark.parametrize("tolerance", [5e-4, 1e-5])
def test_tighter_ewald_tolerances_keep_the_force_contract(Engine, oracle_mod, tune, tolerance):
    """The mixed-precision pair force replaces erfc/exp by a degree-9 polynomial in r^2 only where the fit is good enough: at the
    default ewald tolerance (alpha r_c = 2.15) its residual is 1.2e-6 nm^-3; at 5e-4 (2.63) it is 1.6e-5, at 1e-5 (3.29) 6e-4 -- more
    than the screened force of a pair near the cutoff.  The engine measures the residual of what it fitted and falls back to the
    erfc/exp form: forces of the benchmarked kernel (per-atom lists, a batch member's layout) stay within 1e-5 of the
    oracle at both tolerances (reference blues/simulation.py:219: ewaldErrorTolerance is a user input of the System)."""
    import copy
    base, v = systems.s23k(mobile_atoms=275, frozen=True)
    s = copy.copy(base)
    s.ewald_alpha = float(np.sqrt(-np.log(2.0 * tolerance)) / s.cutoff)
    tune(assume_batch=512)
    g = Engine(s, _integ(20).to_data(precision=0)); g.set_velocities(v)
    assert g.stats()["nonbonded_kernel"] == 2
    mob = np.nonzero(s.mass > 0)[0]
    o = oracle_mod.Oracle(s, _integ(20).to_data(precision=0))
    for ls, le in ((1.0, 1.0), (0.4, 0.2)):
        g.set_global("lambda_sterics", ls); g.set_global("lambda_electrostatics", le)
        eo, fo, _ = o.energy_forces(ls, le)
        f = g.get_forces()[mob]
        assert np.abs(f - fo[mob]).max() <= 1e-5 * np.abs(fo[mob]).max(), (tolerance, np.abs(f - fo[mob]).max() / np.abs(fo[mob]).max())
        assert abs(g.potential_energy() - eo) <= 1e-5 * abs(eo)
    g.step(20)                     # ... and the stepping path (the prune variant of the same pair body) runs
    assert np.isfinite(g.get_global("protocol_work"))
    g.close()


def test_pruned_lists_hold_every_pair_in_range(Engine, oracle_mod, tune):
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    n = 120
    tune(assume_batch=512)                      # the bench's decomposition, engine defaults otherwise (pruned lists on)
    p = Engine(s, _integ(n).to_data(precision=0)); p.set_velocities(v)
    assert p.stats()["pruned_lists"] == 1 and p.stats()["nonbonded_kernel"] == 2
    tune(assume_batch=512, prune_margin=0.0)    # same kernel over the full lists
    q = Engine(s, _integ(n).to_data(precision=0)); q.set_velocities(v)
    assert q.stats()["pruned_lists"] == 0
    mob = np.nonzero(s.mass > 0)[0]
    checked = 0
    for seg in (37, 23, 31):                    # stop at arbitrary points of the lists' lives
        p.step(seg); q.step(seg)
        xp, xq = p.get_positions(), q.get_positions()
        assert np.abs(xp - xq)[mob].max() < 2e-4, np.abs(xp - xq).max()      # (ii) same trajectory up to fp32 summation order, amplified by e^{7.5/ps t}
        fp = p.get_forces()[mob]                # current lists, pruned some steps ago
        r = Engine(s, _integ(n).to_data(precision=0)); r.set_positions(xp)   # fresh lists, no pruning
        for name in ("lambda_sterics", "lambda_electrostatics"):
            r.set_global(name, p.get_global(name))
        fr = r.get_forces()[mob]
        r.close()
        scale = np.abs(fr).max()
        assert np.abs(fp - fr).max() <= 3e-6 * scale, np.abs(fp - fr).max() / scale    # a missing pair at the cutoff would show as ~1e-4
        checked += 1
    o = oracle_mod.Oracle(s, _integ(n).to_data(precision=0)); o.set_positions(p.get_positions())
    fo = o.energy_forces(p.get_global("lambda_sterics"), p.get_global("lambda_electrostatics"))[1][mob]
    fp = p.get_forces()[mob]
    assert np.abs(fp - fo).max() <= 1e-5 * np.abs(fo).max()
    assert np.linalg.norm(fp - fo) <= 1e-5 * np.linalg.norm(fo)
    st = p.stats()
    n_i = int((s.mass > 0).sum()) - len(s.alchemical_atoms)
    assert st["atom_prunes"] > n_i, st     # (iii) a rebuild writes every atom's pruned list itself; in between the atoms asked for more than one prune each
    assert 0 < st["pruned_list_entries"] < 0.85 * st["atom_list_entries"], st
    assert q.stats()["atom_prunes"] == 0
    p.close(); q.close()
    assert checked == 3


def test_list_audit_no_pair_in_range_is_ever_missing(Engine, tune):
    """The engine's own audit (include/blues_engine.h: blues_audit_lists) walks every atom of the system for every atom that has a
    list and counts the pairs inside the cutoff that are entries of NO list the kernel would walk now.  Hot velocities (1.5 x) and
    a hot thermostat make the atoms trip their prune and rebuild triggers often; audited at every step over several list lives
    -- including mobile-mobile pairs, whose two atoms prune at different times."""
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    tune(assume_batch=512)
    g = Engine(s, integrators.generateNCMCIntegrator(nstepsNC=200, dt=0.004, temperature=450.0, seed=11).to_data(precision=0)); g.set_velocities(1.5 * v)
    total = 0
    for step in range(150):
        g.step(1)
        found, missing = g.audit_lists()
        assert missing == 0, (step, found, missing)
        total += found
    st = g.stats()
    assert total > 150 * 50000, total                      # ~ 270 list atoms x a few hundred neighbours each, every step
    assert st["list_builds"] >= 5 and st["atom_prunes"] > 1000, st      # several list lives, many prunes inside each
    g.close()


def test_prune_requests_are_per_chain_inside_a_batch(Engine, tune):
    """Batch = solo bitwise with pruned lists: an atom's list is pruned on ITS OWN displacement flag, so the members of one launch
    prune different atoms at the same step and still reproduce their lone runs bit for bit."""
    from blues_amd.engine import NativeBatch
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    R, n = 8, 40
    tune(assume_batch=R)

    def make():
        out = []
        for r in range(R):
            g = Engine(s, _integ(n, seed=50 + r).to_data(precision=0, replica=r)); g.set_velocities(v * (1.0 + 0.04 * r)); out.append(g)
        return out
    solo = make()
    ws = [g.run_switch(n, trace=True) for g in solo]
    bat = make()
    B = NativeBatch(bat)
    _, wb = B.step(n, trace=True)
    assert B.stats()["fallback_steps"] == 0
    prunes = [g.stats()["atom_prunes"] for g in bat]
    assert len(set(prunes)) > 1, prunes          # the members did not prune in step with each other
    for r in range(R):
        assert np.array_equal(wb[r], ws[r]), r
        assert np.array_equal(solo[r].get_positions(), bat[r].get_positions())
        assert solo[r].stats()["atom_prunes"] == prunes[r]
    for r in (0, R - 1):                         # the audit of a batch member's lists, mid-life
        found, missing = bat[r].audit_lists()
        assert found > 50000 and missing == 0, (r, found, missing)
    B.close()
    for g in solo + bat:
        g.close()
