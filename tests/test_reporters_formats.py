"""On-disk formats and reporters: the AMBER NetCDF layout the reference's NetCDF4Traj writes
(reference blues/formats.py:504-690) with the BLUES per-frame variables, and the reporter scheduling of
reference blues/reporters.py:731-865 / 436-729 -- on the CPU oracle double."""
import io

import numpy as np
import pytest


def _sim(context, tol_box, nsteps=8):
    from blues_amd import integrators, unit
    s, v = tol_box
    integ = integrators.generateNCMCIntegrator(nstepsNC=nsteps, dt=0.002, temperature=300.0, seed=3)
    sim = context.Simulation(None, s, integ)
    sim.context.setVelocities(unit.Quantity(v, "nanometer/picosecond"))
    return sim, s, v


def test_netcdf_layout_and_values(oracle_backed_context, tol_box, tmp_path):
    from scipy.io import netcdf_file
    from blues_amd import reporters
    from blues_amd.formats import AmberNetCDFTraj
    sim, s, v = _sim(oracle_backed_context, tol_box)
    fn = str(tmp_path / "ncmc.nc")
    rep = reporters.NetCDF4Reporter(fn, reportInterval=2, crds=True, vels=True, frcs=True, protocolWork=True, alchemicalLambda=True)
    sim.reporters.append(rep)
    frames_x, frames_w = [], []
    for k in range(4):
        sim.step(2)
        frames_x.append(sim.context._engine.get_positions()); frames_w.append(sim.integrator.get_protocol_work(dimensionless=True))
    rep.close()
    # raw file: the AMBER conventions a reader keys on (formats.py:572-650)
    nc = netcdf_file(fn, "r", mmap=False)
    assert nc.Conventions == b"AMBER" and nc.ConventionVersion == b"1.0"
    assert nc.dimensions["atom"] == s.n_atoms and nc.dimensions["spatial"] == 3 and nc.dimensions["frame"] is None
    var = nc.variables
    assert var["coordinates"].dimensions == ("frame", "atom", "spatial") and var["coordinates"].units == b"angstrom"
    assert var["velocities"].units == b"angstrom/picosecond" and var["velocities"].scale_factor == pytest.approx(20.455)
    assert var["forces"].units == b"kilocalorie/mole/angstrom" and var["time"].units == b"picosecond"
    assert var["cell_lengths"].units == b"angstrom" and var["cell_angles"].units == b"degree"
    assert var["protocolWork"].units == b"kT" and var["alchemicalLambda"].units == b"unitless"
    assert b"".join(var["spatial"][:]) == b"xyz" and b"".join(var["cell_spatial"][:]) == b"abc"
    assert [b"".join(r) for r in var["cell_angular"][:]] == [b"alpha", b"beta ", b"gamma"]
    assert var["coordinates"][:].shape == (4, s.n_atoms, 3)
    assert np.allclose(var["coordinates"][-1], frames_x[-1] * 10.0, atol=1e-4)          # angstrom, float32 on disk
    assert np.allclose(var["cell_lengths"][0], np.asarray(s.box) * 10.0) and np.allclose(var["cell_angles"][0], 90.0)
    nc.close()
    # and through the reader, back in OpenMM units
    t = AmberNetCDFTraj(fn, "r")
    assert t.frame == 4 and t.hasprotocolWork and t.hasalchemicalLambda
    assert np.allclose(t.coordinates[1], frames_x[1], atol=1e-5)
    assert np.allclose(t.velocities[-1], sim.context._engine.get_velocities(), rtol=1e-5, atol=1e-5)
    assert np.allclose(t.protocolWork, frames_w, rtol=1e-6, atol=1e-6)
    assert np.allclose(t.alchemicalLambda, [0.25, 0.5, 0.75, 1.0]) and np.allclose(t.time, [0.004, 0.008, 0.012, 0.016], atol=1e-6)
    kcal = t.forces[-1]
    assert np.isfinite(kcal).all() and np.abs(kcal).max() > 0
    t.close()


def test_frame_indices_cut_the_fused_steps(oracle_backed_context, tol_box, tmp_path):
    """frame_indices=[moveStep, last] (what the reference's config turns 0.5 and -1 into): a frame after step 4 and step 8 of
    an 8-step switch although the driver asks for 4 steps at a time."""
    from blues_amd import reporters
    from blues_amd.formats import AmberNetCDFTraj
    sim, s, v = _sim(oracle_backed_context, tol_box)
    fn = str(tmp_path / "frames.nc")
    rep = reporters.NetCDF4Reporter(fn, frame_indices=[4, 8], crds=True, alchemicalLambda=True)
    assert rep.frame_indices == [3, 7]
    sim.reporters.append(rep)
    assert rep.describeNextReport(sim)[0] == -1
    sim.step(8)                              # one call: cut at the indices
    rep.close()
    t = AmberNetCDFTraj(fn, "r")
    assert t.frame == 2 and np.allclose(t.alchemicalLambda, [0.5, 1.0])
    t.close()


def test_state_data_reporter_lines(oracle_backed_context, tol_box):
    from blues_amd import reporters
    sim, s, v = _sim(oracle_backed_context, tol_box)
    out = io.StringIO()
    rep = reporters.BLUESStateDataReporter(out, reportInterval=4, title="ncmc", step=True, time=True, potentialEnergy=True, kineticEnergy=True,
                                           temperature=True, protocolWork=True, alchemicalLambda=True, currentIter=True, separator="\t")
    sim.reporters.append(rep)
    sim.currentIter = 7
    sim.step(8)
    lines = out.getvalue().strip().split("\n")
    assert lines[0] == '#"Iter"\t"Step"\t"Time (ps)"\t"alchemicalLambda"\t"protocolWork"\t"Potential Energy (kJ/mole)"\t"Kinetic Energy (kJ/mole)"\t"Temperature (K)"'
    assert len(lines) == 3 and all(l.startswith("ncmc: ") for l in lines[1:])
    cols = lines[2][len("ncmc: "):].split("\t")
    assert int(cols[0]) == 7 and int(cols[1]) == 8 and float(cols[3]) == 1.0
    assert float(cols[4]) == pytest.approx(sim.integrator.get_protocol_work(dimensionless=True))
    ndof = 3 * s.n_atoms - len(s.constraint_dist) - 3
    assert float(cols[7]) == pytest.approx(2 * sim.context._engine.kinetic_energy() / (ndof * 0.0083144626), rel=1e-9)
    with pytest.raises(ValueError):
        reporters.BLUESStateDataReporter(out, progress=True)


def test_netcdf_restart_round_trip_and_layout(tmp_path):
    """The Amber restart as the reference writes and reads it: parmed's RestartReporter(outfname + '.rst7', netcdf=True)
    (reference blues/reporters.py:217-225) -> parmed.amber.Rst7 (blues/settings.py:76-85).  Round trip in OpenMM units, the AMBERRESTART
    layout on disk (NetCDF-3 64-bit offsets, no frame dimension, doubles, velocity scale factor 20.455), the ASCII form still read,
    and the reporter mirror writing through a State."""
    from scipy.io import netcdf_file
    from blues_amd import amber, formats, reporters, unit
    rng = np.random.RandomState(4)
    n = 37
    x, v, box = rng.uniform(-1, 5, (n, 3)), rng.normal(0, 0.5, (n, 3)), np.array([4.3572, 6.5358, 8.7144])
    p = str(tmp_path / "restart.rst7")
    amber.write_rst7(p, x, v, box, time_ps=12.5, netcdf=True)
    with open(p, "rb") as fh:
        assert fh.read(4) == b"CDF\x02"                      # classic format, 64-bit offsets
    assert formats.AmberNetCDFRestart.is_netcdf(p)
    x2, v2, box2 = amber.read_rst7(p)                         # dispatches on the magic bytes
    assert np.allclose(x2, x, rtol=0, atol=1e-13) and np.allclose(v2, v, rtol=1e-14, atol=1e-14) and np.allclose(box2, box, atol=1e-13)
    nc = netcdf_file(p, "r", mmap=False)
    try:
        assert nc.Conventions == b"AMBERRESTART" and nc.ConventionVersion == b"1.0"
        assert "frame" not in nc.dimensions and nc.dimensions["atom"] == n and nc.dimensions["spatial"] == 3
        c, w = nc.variables["coordinates"], nc.variables["velocities"]
        assert c.dimensions == ("atom", "spatial") and c.typecode() == "d" and c.units == b"angstrom"
        assert w.units == b"angstrom/picosecond" and w.scale_factor == pytest.approx(20.455)
        assert np.allclose(np.array(c[:]), x * 10.0) and np.allclose(np.array(w[:]) * 20.455, v * 10.0)      # angstrom; Amber's time unit
        assert nc.variables["time"].units == b"picosecond" and float(nc.variables["time"].getValue()) == 12.5
        assert np.allclose(np.array(nc.variables["cell_lengths"][:]), box * 10.0) and np.allclose(np.array(nc.variables["cell_angles"][:]), 90.0)
        assert bytes(np.array(nc.variables["cell_angular"][:]).tobytes()) == b"alphabeta gamma"
    finally:
        nc.close()
    # a NetCDF file that is not a restart is refused, the ASCII form is still read
    traj = str(tmp_path / "traj.nc")
    t = formats.AmberNetCDFTraj.open_new(traj, n, True); t.add_coordinates(x); t.add_cell_lengths_angles(box); t.add_time(0.0); t.close()
    with pytest.raises(ValueError, match="AMBERRESTART"):
        formats.AmberNetCDFRestart.read(traj)
    pa = str(tmp_path / "ascii.rst7")
    amber.write_rst7(pa, x, v, box)
    xa, va, ba = amber.read_rst7(pa)
    assert np.allclose(xa, x, atol=1e-7) and np.allclose(va, v, atol=1e-5) and np.allclose(ba, box, atol=1e-7)

    class _State:
        def getPositions(self, asNumpy=False): return unit.Quantity(x, "nanometer")
        def getVelocities(self, asNumpy=False): return unit.Quantity(v, "nanometer/picosecond")
        def getPeriodicBoxVectors(self, asNumpy=False): return unit.Quantity(np.diag(box), "nanometer")
        def getTime(self): return unit.Quantity(3.0, "picosecond")

    class _Sim:
        currentStep = 40
    rep = reporters.RestartReporter(str(tmp_path / "out.rst7"), reportInterval=20, netcdf=True)
    assert rep.describeNextReport(_Sim())[:3] == (20, True, True)
    rep.report(_Sim(), _State())
    x3, v3, b3 = amber.read_rst7(str(tmp_path / "out.rst7"))
    assert np.allclose(x3, x, atol=1e-13) and np.allclose(v3, v, atol=1e-13) and np.allclose(b3, box, atol=1e-13)


def test_enforce_periodic_box_wraps_whole_molecules(oracle_backed_context, tol_box):
    """getState(enforcePeriodicBox=True) (reference blues/simulation.py:874-881): every molecule's centre inside the box, molecules whole,
    lattice translations only; a Move's positions[atom_indices] sees the same wrapped coordinates as the full array."""
    from blues_amd import integrators, unit
    from blues_amd.context import PeriodicWrapper
    s, v = tol_box
    box = np.asarray(s.box)
    sim = oracle_backed_context.Simulation(None, s, integrators.generateNCMCIntegrator(nstepsNC=4, dt=0.002, seed=1))
    x = np.array(s.positions, dtype=float)
    x[:15] += box * np.array([2, -1, 0])            # the ligand two boxes away; one water one box away
    x[15:18] -= box * np.array([0, 0, 1])
    sim.context.setPositions(unit.Quantity(x, "nanometer"))
    raw = sim.context.getState(getPositions=True).getPositions(asNumpy=True)._value
    wrapped = sim.context.getState(getPositions=True, enforcePeriodicBox=True).getPositions(asNumpy=True)._value
    w = PeriodicWrapper(s)
    assert len(np.unique(w.molecule_of)) == 1 + (s.n_atoms - 15) // 3 and len(set(w.molecule_of[:15])) == 1
    d = (wrapped - raw) / box
    assert np.allclose(d, np.round(d), atol=1e-12)                                       # lattice translations only
    for m in range(w.molecule_of.max() + 1):
        a = w.atoms_of(m)
        assert len(np.unique(np.round(d[a]), axis=0)) == 1                               # the same one for a whole molecule
        c = wrapped[a].mean(0)
        assert np.all(c >= -1e-12) and np.all(c < box + 1e-12)                           # centre inside the box
    sh = w.shifts_for(list(range(15)), lambda need: raw[np.asarray(need)])
    assert np.allclose(raw[:15] + sh, wrapped[:15], atol=1e-12)
