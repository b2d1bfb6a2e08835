"""GPU suite: the fragment-list nonbonded path (k1_mode 3, blues_amd/csrc/kernels_frag.h) -- the force path of systems whose
environment moves: the MD leg of a BLUES iteration (reference blues/simulation.py:1189-1213, the unfrozen MD System), switches without
freeze_radius (reference examples/example_water.py) and NCMC Systems whose mobile atoms have scattered over MD legs (reference
blues/simulation.py:1028-1037 hands such a State over every iteration).

Bar (north_star): energies, forces, protocol work within 1e-5 of the fp64 CPU restatement; the lists hold every pair inside the
cutoff at every step (blues_audit_lists); a batch member equals the same chain advanced alone, bit for bit.
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


def _integ(n, seed=11):
    return integrators.generateNCMCIntegrator(nstepsNC=n, dt=0.002, temperature=300.0, seed=seed)


def _rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def _variants(tol_box):
    """(name, System): the alchemical toluene in water with nothing frozen; the same System without an alchemical group (the MD
    System: the toluene's exclusions and 1-4 pairs go through the fragment masks); two thirds of the waters frozen."""
    s, v = tol_box
    md = copy.copy(s); md.alchemical_atoms = np.zeros(0, np.int32)
    res = np.asarray(s.residue_of_atom)
    frozen = copy.copy(s)
    keep = np.nonzero((res % 3 == 0) | (np.arange(s.n_atoms) < 15))[0]
    frozen = systems.freeze_except(frozen, keep)
    return [("alchemical", s), ("md", md), ("partly_frozen", frozen)]


@pytest.mark.parametrize("which", [0, 1, 2])
def test_energies_and_forces_match_the_oracle(Engine, oracle_mod, tol_box, tune, which):
    name, s = _variants(tol_box)[which]
    tune(k1_mode=3)
    data = _integ(20).to_data(precision=0)
    g, o = Engine(s, data), oracle_mod.Oracle(s, data)
    assert g.stats()["nonbonded_kernel"] == 3, g.stats()
    mob = s.mass > 0
    for ls, le in ((1.0, 1.0), (0.3, 0.0)):
        eo, fo, to = o.energy_forces(ls, le)
        g.set_global("lambda_sterics", ls); g.set_global("lambda_electrostatics", le)
        tg = g.energy_terms()
        assert abs(tg.sum() - eo) <= 1e-5 * abs(eo), (name, tg.sum(), eo)
        for k in range(8):
            assert abs(tg[k] - to[k]) <= 1e-5 * max(abs(to[k]), 1.0, 1e-3 * abs(eo)), (name, k, tg[k], to[k])
        fg = g.get_forces()
        assert _rel(fg[mob], fo[mob]) <= 1e-5, (name, _rel(fg[mob], fo[mob]))
        assert np.all(fg[~mob] == 0.0)
    found, missing = g.audit_lists()
    assert found > 0 and missing == 0
    g.close()


@pytest.mark.parametrize("which", [0, 2])
def test_switch_work_and_list_audit(Engine, oracle_mod, tol_box, tune, which):
    """A 40-step switch against the oracle; the lists are audited as they age (rebuilds of the outer lists, prunes of the inner
    ones: both happen within these steps at this box's margins)."""
    name, s = _variants(tol_box)[which]
    v = tol_box[1].copy(); v[s.mass == 0.0] = 0.0
    tune(k1_mode=3)
    n = 40
    data = _integ(n).to_data(precision=0)
    g, o = Engine(s, data), oracle_mod.Oracle(s, data)
    g.set_velocities(v); o.set_velocities(v)
    wg = []
    for seg in range(4):
        wg.extend(g.run_switch(10, trace=True))
        found, missing = g.audit_lists()
        assert found > 0 and missing == 0, (name, seg, found, missing)
    wo = []
    for _ in range(n):
        o.step(1); wo.append(o.get_global("protocol_work"))
    wg, wo = np.array(wg), np.array(wo)
    assert np.abs(wg - wo).max() <= 1e-5 * np.abs(wo).max(), (name, np.abs(wg - wo).max() / np.abs(wo).max())
    st = g.stats()
    assert st["nonbonded_kernel"] == 3 and st["pruned_lists"] == 1 and st["list_builds"] >= 1
    g.close()


def test_md_leg_through_fragment_lists(Engine, oracle_mod, tol_box, tune):
    """The MD leg (reference blues/simulation.py:1189-1213, openmm.LangevinIntegrator) on the non-alchemical System."""
    s, v = tol_box
    md = copy.copy(s); md.alchemical_atoms = np.zeros(0, np.int32)
    tune(k1_mode=3)
    data = integrators.LangevinIntegrator(300.0, 1.0, 0.002, seed=5).to_data(precision=0)
    g, o = Engine(md, data), oracle_mod.Oracle(md, data)
    g.set_velocities(v); o.set_velocities(v)
    g.step(30); o.step(30)
    assert g.stats()["nonbonded_kernel"] == 3
    assert np.abs(g.get_positions() - o.get_positions()).max() < 2e-4
    assert abs(g.potential_energy() - o.potential_energy()) <= 2e-5 * abs(o.potential_energy())
    assert g.audit_lists()[1] == 0
    g.close()


def test_batch_member_equals_the_lone_chain(Engine, tol_box, tune):
    from blues_amd.engine import NativeBatch
    s, v = tol_box
    R, n = 4, 60
    tune(k1_mode=3, assume_batch=R)

    def make(r):
        g = Engine(s, _integ(n, seed=100 + r).to_data(precision=0, replica=r)); g.set_velocities(v * (1.0 + 0.01 * r)); return g
    alone = [make(r) for r in range(R)]
    for g in alone:
        g.step(n)
    together = [make(r) for r in range(R)]
    B = NativeBatch(together)
    B.step(n)
    for a, b in zip(alone, together):
        assert b.stats()["nonbonded_kernel"] == 3
        assert np.array_equal(a.get_positions(), b.get_positions()) and np.array_equal(a.get_velocities(), b.get_velocities())
        assert a.get_global("protocol_work") == b.get_global("protocol_work")
    assert B.stats()["lockstep_steps"] >= n
    B.close()
    for g in alone + together:
        g.close()


def test_scattered_mobile_atoms_fall_back_to_fragment_lists(Engine, tune):
    """An NCMC System with freeze_radius keeps its mobile atoms by INDEX (reference blues/simulation.py:394-480): after MD legs they
    are scattered through the box, and the per-atom lists over group images (sized for tiles of neighbouring mobile atoms) cannot hold
    them.  The engine then lays itself out with fragment lists; energies agree with a freshly built engine on the same coordinates."""
    s, v = systems.s23k(mobile_atoms=275, frozen=True)
    tune(assume_batch=64)
    g = Engine(s, _integ(10).to_data(precision=0))
    assert g.stats()["nonbonded_kernel"] == 2
    e0 = g.potential_energy()
    # scatter the mobile waters: exchange the coordinates of mobile and frozen waters far apart (whole molecules, same geometry)
    x = s.positions.copy()
    res = np.asarray(s.residue_of_atom)
    mob_res = np.unique(res[(s.mass > 0) & (np.arange(s.n_atoms) >= 15)])
    frozen_res = np.unique(res[(s.mass == 0)])
    rng = np.random.RandomState(3)
    partners = rng.choice(frozen_res[(frozen_res % 24) != 0], size=len(mob_res), replace=False)
    for a, b in zip(mob_res, partners):
        ia, ib = np.nonzero(res == a)[0], np.nonzero(res == b)[0]
        if len(ia) == 3 and len(ib) == 3:
            x[ia], x[ib] = x[ib].copy(), x[ia].copy()
    g.set_positions(x)
    e1 = g.potential_energy()
    assert g.stats()["nonbonded_kernel"] == 3, g.stats()
    assert abs(e1 - e0) <= 1e-6 * abs(e0)      # (identical molecules exchanged places: the same configuration)
    g.set_velocities(v)
    g.step(10)
    assert g.audit_lists()[1] == 0
    g.close()


def test_resort_by_age_keeps_a_member_equal_to_the_lone_chain(Engine, tol_box, tune):
    """Fragment-list engines of an all-mobile system re-sort by AGE (4096 steps in one order; blues_engine.hip: resort_by_age): a
    function of the chain's own step count, so the members of a batch re-sort together (the host's cores share them) and still do
    exactly what the lone chain does.  4,200 steps cross that age once."""
    from blues_amd.engine import NativeBatch
    s, v = tol_box
    md = copy.copy(s); md.alchemical_atoms = np.zeros(0, np.int32)
    R, n = 3, 4200
    tune(k1_mode=3, assume_batch=R)

    def make(r):
        g = Engine(md, integrators.LangevinIntegrator(300.0, 1.0, 0.002, seed=40 + r).to_data(precision=0, replica=r)); g.set_velocities(v * (1.0 + 0.01 * r)); return g
    alone = [make(r) for r in range(R)]
    for g in alone:
        g.step(n)
    together = [make(r) for r in range(R)]
    B = NativeBatch(together)
    B.step(n)
    for a, b in zip(alone, together):
        assert a.stats()["resorts"] >= 1 and b.stats()["resorts"] == a.stats()["resorts"]
        assert np.array_equal(a.get_positions(), b.get_positions()) and np.array_equal(a.get_velocities(), b.get_velocities())
        assert b.audit_lists()[1] == 0
    B.close()
    for g in alone + together:
        g.close()


def test_fragment_lists_get_more_room_before_a_row_overflows(Engine, tune):
    """A row of the fragment lists that reaches its last chunk asks for a new layout, and the host then gives the lists a quarter more room
    (blues_engine.hip: frag_grow_caps) -- before an entry is dropped.  The capacities are shrunk (acap_scale) until the first list build
    overflows; one 64-entry step above that, the longest row sits in its last chunk: the engine must re-lay itself out at its first
    poll and carry on with clean lists.  An overflow names the capacity that was exceeded."""
    from blues_amd.engine import EngineError
    s, v = systems.s23k(frozen=False, restrained=0)
    s = copy.copy(s); s.alchemical_atoms = np.zeros(0, np.int32)
    data = integrators.LangevinIntegrator(300.0, 1.0, 0.004, seed=9).to_data(precision=0)

    def attempt(scale, nsteps):
        tune(k1_mode=3, acap_scale=scale)
        g = Engine(s, data)
        try:
            g.set_velocities(v)
            g.step(nsteps)
            return g.stats(), g.audit_lists(), None
        except EngineError as e:
            return None, None, str(e)
        finally:
            g.close()

    last_ok = None
    scale = 1.0
    while scale > 0.4:
        st, au, err = attempt(scale, 2)
        if err is not None:
            assert "fragment lists" in err and ("outer rows" in err or "inner rows" in err), err
            break
        last_ok = scale
        scale -= 0.04
    assert last_ok is not None and scale > 0.4, "no capacity was small enough to overflow"
    st, au, err = attempt(last_ok, 200)
    if err is not None:   # (the longest row grew past the capacity in the 64 steps before the first poll)
        assert "fragment lists" in err
        pytest.skip("the row outgrew its capacity before the first poll: " + err)
    assert st["resorts"] >= 1 and st["nonbonded_kernel"] == 3, st
    assert au[0] > 0 and au[1] == 0
