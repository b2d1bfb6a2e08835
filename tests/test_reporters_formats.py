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
