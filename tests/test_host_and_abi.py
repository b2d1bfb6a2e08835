"""CPU suite, part 2: the C-ABI library (loads, exports every declared symbol, fails loudly without a
GPU), host-side parsing, and the BLUES driver mirror run against an oracle-backed test double."""
import ctypes
import os
import re

import numpy as np
import pytest

from blues_amd import _abi, amber, build, integrators, lepton, moves, simulation, systems, unit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_hipcc():
    try:
        build.hipcc_path(); return True
    except RuntimeError:
        return False


@pytest.fixture(scope="module")
def engine_lib():
    if not os.path.exists(build.LIB_PATH) or build.is_stale():
        if not _have_hipcc():
            pytest.skip("hipcc not available to build the engine")
        build.build_engine()
    return ctypes.CDLL(build.LIB_PATH)


def test_library_exports_every_declared_symbol(engine_lib):
    header = open(os.path.join(ROOT, "include", "blues_engine.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char \*)\s*\*?(blues_[a-z_]+)\(", header, flags=re.M))
    assert declared == set(_abi.ENGINE_SYMBOLS), declared ^ set(_abi.ENGINE_SYMBOLS)
    for name in declared:
        assert hasattr(engine_lib, name), name
    _abi.declare_engine_prototypes(engine_lib)
    assert engine_lib.blues_abi_version() == _abi.ABI_VERSION


def test_tuning_struct_mirror_and_no_environment_reads(engine_lib):
    """BluesTuning: the ctypes mirror has the header's fields in the header's order, the defaults round-trip, and the native
    sources read no environment variable (the launch policy is an explicit, documented struct)."""
    header = open(os.path.join(ROOT, "include", "blues_engine.h")).read()
    body = re.search(r"typedef struct BluesTuning \{(.*?)\} BluesTuning;", header, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(int32_t|double)\s+([a-z_0-9]+);", body)
    ctype = {"int32_t": ctypes.c_int32, "double": ctypes.c_double}
    assert [(n, ctype[t]) for t, n in fields] == list(_abi.BluesTuning._fields_)
    _abi.declare_engine_prototypes(engine_lib)
    t = _abi.BluesTuning()
    engine_lib.blues_tuning_default(t)
    assert t.struct_size == ctypes.sizeof(_abi.BluesTuning) and t.fork == 1 and t.prune_margin < 0 and t.skin == 0.0
    t.skin = 0.2; t.fork = 0
    assert engine_lib.blues_set_tuning(t) == 0
    u = _abi.BluesTuning(); engine_lib.blues_get_tuning(u)
    assert u.skin == 0.2 and u.fork == 0
    t.struct_size = 8
    assert engine_lib.blues_set_tuning(t) != 0          # a caller built against another layout is refused
    assert engine_lib.blues_set_tuning(None) == 0
    engine_lib.blues_get_tuning(u)
    assert u.skin == 0.0 and u.fork == 1
    for f in os.listdir(build.CSRC):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(build.CSRC, f)).read(), f


def test_engine_fails_loudly_without_gpu(tol_box):
    from conftest import gpu_available
    if gpu_available():
        pytest.skip("a GPU is present")
    from blues_amd.engine import EngineError, NativeEngine
    s, _ = tol_box
    with pytest.raises(EngineError, match="no CPU fallback|HIP"):
        NativeEngine(s, integrators.generateNCMCIntegrator(nstepsNC=10).to_data())


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "blues_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                code = [l for l in open(os.path.join(dirpath, f)).read().split("\n")
                        if re.match(r"\s*(from|import|#include)\b", l) or "CDLL" in l or "dlopen" in l]
                for l in code:
                    assert "oracle" not in l, (f, l)


def test_descriptor_roundtrip(tol_box):
    s, _ = tol_box
    d, keep = s.to_desc()
    assert d.n_atoms == 975 and d.n_constraints == len(s.constraint_dist) and d.n_alchemical == 15
    assert d.box[0] == pytest.approx(2.1786) and d.box[1] == 0.0
    assert d.nonbonded_method == _abi.NB_PME_DIRECT and d.ewald_alpha == pytest.approx(2.145966, abs=1e-6)
    i = integrators.generateNCMCIntegrator(nstepsNC=100).to_data()
    di, _ = i.to_desc()
    assert di.n_lambda_steps == 200 and di.splitting == b"H V R O R V H" and di.collision_rate == 1.0
    with pytest.raises(ValueError):
        _abi.IntegratorData(timestep=0.002, temperature=300, nsteps_neq=10, lambda_sterics=[1, 1], lambda_electrostatics=[1, 1]).to_desc()


def test_lepton_subset():
    f = lepton.compile_expression("2*lambda^2 + step(lambda-0.5) - min(1, max(0.1, lambda)) + abs(-3) + select(lambda, 1, 2)")
    assert f(**{"lambda": 0.5}) == pytest.approx(2 * 0.25 + 1 - 0.5 + 3 + 1)
    assert lepton.compile_expression("1")(**{"lambda": 0.3}) == 1.0
    for bad in ("__import__('os')", "lambda.real", "foo(1)", "x+1"):
        with pytest.raises((ValueError, SyntaxError)):
            lepton.compile_expression(bad)


def test_unit_shim():
    kT = unit.Quantity(2.494339, "kilojoule/mole")
    e = unit.Quantity(10.0, "kilojoule/mole")
    assert (e - e + e) * (-1.0 / kT) == pytest.approx(-10.0 / 2.494339)
    assert unit.Quantity(10.0, "angstroms").value_in_unit("nanometers") == pytest.approx(1.0)
    assert unit.Quantity(1.0, "kilocalorie/mole").value_in_unit("kilojoule/mole") == pytest.approx(4.184)
    assert unit.value_in(unit.Quantity(4.0, "femtoseconds"), "picosecond") == pytest.approx(0.004)
    with pytest.raises(TypeError):
        unit.Quantity(1.0, "kelvin").value_in_unit("nanometer")


@pytest.mark.skipif(not os.path.exists("/root/reference/blues/tests/data/TOL-parm.prmtop"), reason="reference checkout not present")
def test_amber_reader_on_reference_fixture(known_answers, tol_box):
    prm = amber.read_prmtop("/root/reference/blues/tests/data/TOL-parm.prmtop")
    p = known_answers["tol_parm_pointers"]
    assert int(prm["POINTERS"][0]) == p["NATOM"] and int(prm["POINTERS"][1]) == p["NTYPES"]
    assert int(prm["POINTERS"][2]) == p["NBONH"] and int(prm["POINTERS"][10]) == p["NNB"] and int(prm["POINTERS"][11]) == p["NRES"]
    pos, vel, box = amber.read_inpcrd("/root/reference/blues/tests/data/TOL-parm.inpcrd")
    assert pos.shape == (975, 3) and np.allclose(box, 2.1786)
    s = amber.system_from_amber(prm, pos, box, hydrogen_mass=3.024, alchemical_atoms=range(15))
    ref, _ = tol_box
    assert np.allclose(s.charge, ref.charge) and np.allclose(s.sigma, ref.sigma) and np.allclose(s.epsilon, ref.epsilon)
    assert np.array_equal(s.exclusions, ref.exclusions) and np.allclose(s.exception_params, ref.exception_params)
    # toluene charges (e), reference TOL-parm.prmtop:64-66 / 18.2223
    assert np.allclose(s.charge[:3], [-0.0538, -0.0773, -0.131], atol=5e-5) and np.allclose(s.charge[15:18], [-0.834, 0.417, 0.417], atol=1e-6)
    # createSystem(nonbondedMethod=PME) in full by default: the mesh of OpenMM's Reference platform at ewaldErrorTolerance 0.005
    from blues_amd._abi import NB_PME, NB_PME_DIRECT
    assert s.nonbonded_method == NB_PME and tuple(s.pme_grid) == (9, 9, 9) and s.pme_order == 5 and s.dispersion_correction
    assert amber.system_from_amber(prm, pos, box, reciprocal_space=False).nonbonded_method == NB_PME_DIRECT


def test_s23k_construction():
    big, v = systems.s23k(mobile_atoms=275)
    assert big.n_atoms == 23400 and np.allclose(big.box, [4.3572, 6.5358, 8.7144])
    n_mobile = int((big.mass > 0).sum())
    assert 270 <= n_mobile <= 280 and (big.mass[:15] > 0).all()
    assert abs(big.charge.sum()) < 1e-4 and len(big.alchemical_atoms) == 15
    assert np.all(v[big.mass == 0] == 0)
    # whole molecules only: every constraint joins two mobile or two frozen atoms
    c = big.constraint_atoms
    assert np.all((big.mass[c[:, 0]] > 0) == (big.mass[c[:, 1]] > 0))
    full, _ = systems.s23k(frozen=False, restrained=50)
    assert (full.mass > 0).all() and len(full.restraint_atoms) == 50 and full.restraint_k == pytest.approx(2092.0)


def test_freeze_radius_semantics(tol_box):
    s, _ = tol_box
    water = np.arange(s.n_atoms) >= 15
    fr = systems.freeze_radius(s, np.arange(15), 0.5, solvent_mask=water)
    assert (fr.mass[:15] > 0).all() and (fr.mass[15:] == 0).all()      # solvent always frozen
    fr2 = systems.freeze_radius(s, np.arange(15), 0.5)
    mob = fr2.mass > 0
    assert mob[:15].all() and 15 < mob.sum() < s.n_atoms
    res = s.residue_of_atom
    for r in np.unique(res[mob]):
        assert mob[res == r].all()                                      # whole residues


class _CountingMove(moves.Move):
    def __init__(self): self.calls = []
    def beforeMove(self, ctx): self.calls.append(("before", ctx._integrator.getGlobalVariableByName("step"))); return ctx
    def move(self, ctx): self.calls.append(("move", ctx._integrator.getGlobalVariableByName("step"))); return ctx
    def afterMove(self, ctx): self.calls.append(("after", ctx._integrator.getGlobalVariableByName("step"))); return ctx


def _make_sim(context_mod, s, v, nsteps=8, seed=5):
    integ = integrators.generateNCMCIntegrator(nstepsNC=nsteps, dt=0.002, temperature=300.0, seed=seed)
    sim = context_mod.Simulation(None, s, integ)
    sim.context.setVelocities(unit.Quantity(v, "nanometer/picosecond"))
    return sim


def test_driver_hook_order_and_state_table(oracle_backed_context, tol_box):
    s, v = tol_box
    sim = _make_sim(oracle_backed_context, s, v)
    mv = _CountingMove()
    b = simulation.BLUESSimulation(simulation.SimulationSet(sim), {"nstepsNC": 8, "moveStep": 4, "nIter": 1}, moves.MoveEngine(mv))
    b._syncStatesMDtoNCMC()
    x_before = sim.context.getState(getPositions=True).getPositions(asNumpy=True)._value.copy()
    b._stepNCMC(8, 4)
    assert mv.calls == [("before", 0.0), ("move", 4.0), ("after", 8.0)]  # reference blues/simulation.py:1066-1086
    st0, st1 = b.stateTable["ncmc"]["state0"], b.stateTable["ncmc"]["state1"]
    for key in ("positions", "velocities", "potential_energy", "kinetic_energy", "box_vectors"):
        assert key in st0 and key in st1
    assert np.not_equal(st0["positions"]._value, st1["positions"]._value).all()  # reference blues/tests/test_simulation.py:385-391
    assert isinstance(b._computeAlchemicalCorrection(), float)                   # test_simulation.py:393-395
    # direct-space-only model: U_md == U_ncmc(lambda=1) so the correction vanishes when the protocol ends at lambda=1
    assert abs(b._computeAlchemicalCorrection()) < 1e-9
    b._acceptRejectMove()
    assert b.accept + b.reject == 1 and set(b.last) >= {"accept", "log_accept", "protocol_work"}
    x_after = sim.context.getState(getPositions=True).getPositions(asNumpy=True)._value
    if b.last["accept"]:
        d = x_after - st1["positions"]._value     # (state1 was taken with enforcePeriodicBox=True: equal up to whole box vectors per molecule)
        assert np.allclose(d - np.asarray(s.box) * np.round(d / np.asarray(s.box)), 0.0, atol=1e-12)
    else:
        d = x_after - x_before                     # reject restores the pre-switch state (state0 was taken with enforcePeriodicBox=True: up to whole box vectors)
        assert np.allclose(d - np.asarray(s.box) * np.round(d / np.asarray(s.box)), 0.0, atol=1e-12)
    b._resetSimulations(300.0)
    integ = sim.context._integrator
    assert integ.getGlobalVariableByName("step") == 0 and integ.getGlobalVariableByName("protocol_work") == 0 and sim.currentStep == 0
    assert integ.getLogAcceptanceProbability(sim.context) == 0.0


def test_driver_error_policy(oracle_backed_context, tol_box):
    """Any exception in a step aborts the switch after move._error (reference blues/simulation.py:1088-1094)."""
    s, v = tol_box
    sim = _make_sim(oracle_backed_context, s, v)

    class Boom(moves.Move):
        errored = False
        def move(self, ctx): raise RuntimeError("Particle coordinate is nan")
        def _error(self, ctx): Boom.errored = True; return ctx

    eng = moves.MoveEngine(Boom())
    b = simulation.BLUESSimulation(simulation.SimulationSet(sim), {"nstepsNC": 8, "moveStep": 4, "nIter": 1}, eng)
    b._syncStatesMDtoNCMC()
    with pytest.raises(SystemExit):  # MoveEngine.runEngine turns a broken move into SystemExit (reference blues/moves.py:400-408)
        b._stepNCMC(8, 4)


def test_rotation_move_and_full_run(oracle_backed_context, tol_box):
    s, v = tol_box
    sim = _make_sim(oracle_backed_context, s, v, nsteps=6, seed=9)
    lig = np.arange(15)
    mv = moves.RandomLigandRotationMove(lig, s.mass[lig], random_state=3)
    x0 = sim.context.getState(getPositions=True).getPositions(asNumpy=True)._value.copy()
    mv.move(sim.context)
    x1 = sim.context.getState(getPositions=True).getPositions(asNumpy=True)._value
    assert np.not_equal(x0[lig], x1[lig]).all() and np.array_equal(x0[15:], x1[15:])  # reference blues/tests/test_randomrotation.py:52-61
    d0 = np.linalg.norm(x0[0] - x0[5]); d1 = np.linalg.norm(x1[0] - x1[5])
    assert d0 == pytest.approx(d1, rel=1e-6)                                            # rigid rotation
    com = lambda x: (x[lig] * s.mass[lig, None]).sum(0) / s.mass[lig].sum()
    assert np.allclose(com(x0), com(x1), atol=1e-6)
    R = moves.rotation_matrix_from_quaternion(moves.uniform_quaternion(1))
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and np.linalg.det(R) == pytest.approx(1.0)
    np.random.seed(0)
    b = simulation.BLUESSimulation(simulation.SimulationSet(sim), {"nstepsNC": 6, "moveStep": 3, "nIter": 2}, moves.MoveEngine(mv))
    b.run()
    assert b.accept + b.reject == 2 and b.acceptRatio == b.accept / 2.0


def test_simulation_step_reporters(oracle_backed_context, tol_box):
    s, v = tol_box
    sim = _make_sim(oracle_backed_context, s, v, nsteps=10)

    class Rep:
        def __init__(self): self.seen = []
        def describeNextReport(self, simulation):
            return (4 - simulation.currentStep % 4, True, False, False, True)
        def report(self, simulation, state):
            self.seen.append((simulation.currentStep, state.getPotentialEnergy()._value, simulation.integrator.get_protocol_work(dimensionless=True)))

    rep = Rep(); sim.reporters.append(rep)
    sim.step(10)
    assert [r[0] for r in rep.seen] == [4, 8] and sim.currentStep == 10
    assert sim.integrator.getGlobalVariableByName("lambda") == pytest.approx(1.0)
    with pytest.raises(Exception, match="does not contain"):
        sim.context.getState(getPositions=True).getVelocities()
    # positional form used by reference blues/moves.py:1218
    st = sim.context.getState(True, True, False, True, True, False)
    assert st.getPositions(asNumpy=True)._value.shape == (975, 3) and "lambda_sterics" in st.getParameters()


def test_header_is_plain_c(tmp_path):
    """include/blues_engine.h must be consumable from C (the boundary is a C-ABI, not C++): compile a C99 translation
    unit that uses every entry point's prototype, and link it against the engine library when that exists."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    src = tmp_path / "abi_user.c"
    calls = "\n".join("    (void)&%s;" % name for name in _abi.ENGINE_SYMBOLS)
    src.write_text('#include "blues_engine.h"\n#include <stddef.h>\nint main(void) {\n    BluesSystemDesc s; BluesIntegratorDesc i; BluesEngine *h = NULL;\n'
                   '    (void)s; (void)i; (void)h;\n%s\n    return blues_abi_version() == BLUES_ABI_VERSION ? 0 : 1;\n}\n' % calls)
    obj = tmp_path / "abi_user.o"
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(obj)])
    # struct layout agreed between C and the ctypes mirror
    probe = tmp_path / "sizes.c"
    probe.write_text('#include "blues_engine.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(void) { printf("%zu %zu %zu %zu\\n", sizeof(BluesSystemDesc), sizeof(BluesIntegratorDesc), '
                     'offsetof(BluesSystemDesc, remove_cm_motion), offsetof(BluesIntegratorDesc, precision)); return 0; }\n')
    exe = tmp_path / "sizes"
    subprocess.check_call([gcc, "-std=c99", "-I", os.path.join(ROOT, "include"), str(probe), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).split()
    assert int(out[0]) == ctypes.sizeof(_abi.BluesSystemDesc) and int(out[1]) == ctypes.sizeof(_abi.BluesIntegratorDesc)
    assert int(out[2]) == _abi.BluesSystemDesc.remove_cm_motion.offset and int(out[3]) == _abi.BluesIntegratorDesc.precision.offset


def test_amber_restart_round_trip(tmp_path):
    """Amber ASCII restart files (.rst7: the inpcrd layout plus velocities; reference blues/settings.py:60-90 passes them to ParmEd):
    what write_rst7 writes, read_rst7 / read_inpcrd read back to the 7 decimals of the format, with and without velocities and box."""
    from blues_amd import amber
    rng = np.random.RandomState(4)
    x = rng.uniform(-1.0, 5.0, (7, 3)); v = rng.normal(0.0, 0.5, (7, 3)); box = np.array([4.3572, 6.5358, 8.7144])
    p = str(tmp_path / "a.rst7")
    amber.write_rst7(p, x, v, box)
    x2, v2, b2 = amber.read_rst7(p)
    assert np.abs(x2 - x).max() < 6e-9 and np.abs(v2 - v).max() < 2e-7 and np.abs(b2 - box).max() < 6e-9
    amber.write_rst7(p, x, None, box)
    x3, v3, b3 = amber.read_inpcrd(p)
    assert v3 is None and np.abs(x3 - x).max() < 6e-9 and np.abs(b3 - box).max() < 6e-9
    amber.write_rst7(p, x[:2], None, None)       # two atoms: a single, partly filled coordinate line and nothing else
    x4, v4, b4 = amber.read_rst7(p)
    assert x4.shape == (2, 3) and v4 is None and b4 is None


def test_build_in_parallel_keeps_order_and_raises():
    """replicas.build_in_parallel: the chains of a GPU are created on host threads (their set-up is native host work); the result
    is in chain order whatever thread made it, and the first failure surfaces as it would from the plain loop."""
    import threading
    import time
    from blues_amd.replicas import build_in_parallel
    seen = set()

    def make(i):
        time.sleep(0.002 * ((7 * i) % 5))
        seen.add(threading.get_ident())
        return i * i
    assert build_in_parallel(make, 40, workers=8) == [i * i for i in range(40)]
    assert len(seen) > 1
    assert build_in_parallel(make, 5, workers=1) == [0, 1, 4, 9, 16]
    assert build_in_parallel(make, 0) == []

    def bad(i):
        if i == 3:
            raise ValueError("chain 3")
        return i
    with pytest.raises(ValueError, match="chain 3"):
        build_in_parallel(bad, 8, workers=4)
