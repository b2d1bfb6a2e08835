"""On-disk trajectory format of the NCMC / MD legs: AMBER NetCDF with the two per-frame variables BLUES adds
(`protocolWork` in kT, `alchemicalLambda`), as written by the reference's NetCDF4Traj
(reference blues/formats.py:476-690, on top of parmed.amber.netcdffiles.NetCDFTraj).

The reference needs ParmEd + netCDF4; this writer/reader uses scipy.io.netcdf_file (NetCDF-3, 64-bit offsets -- the
classic AMBER trajectory container, which is also what ParmEd falls back to without netCDF4) and nothing else, so it
runs on the GPU box.  Units on disk are AMBER's: angstrom, angstrom/picosecond / 20.455, kilocalorie/mole/angstrom,
degrees, picoseconds."""
import numpy as np

AMBER_VELOCITY_SCALE = 20.455     # AMBER's internal time unit (1/20.455 ps): stored velocity * scale_factor = angstrom/ps
_NM_TO_A = 10.0
_KJ_PER_NM_TO_KCAL_PER_A = 1.0 / (4.184 * 10.0)


class AmberNetCDFTraj(object):
    """One AMBER NetCDF trajectory file.  `open_new` lays the file out (which per-frame variables exist is fixed then);
    each `add_*` appends to its own frame counter, exactly one call per frame and variable (reference formats.py:659-690)."""

    def __init__(self, fname, mode="r"):
        from scipy.io import netcdf_file
        self.fname = fname
        self._nc = netcdf_file(fname, mode, version=2, mmap=False)
        self._frames = {}
        if mode == "r":
            v = self._nc.variables
            self.hascrds, self.hasvels, self.hasfrcs = "coordinates" in v, "velocities" in v, "forces" in v
            self.hasbox = "cell_lengths" in v
            self.hasprotocolWork, self.hasalchemicalLambda = "protocolWork" in v, "alchemicalLambda" in v
            self.atom = self._nc.dimensions["atom"]
            self.frame = len(v["time"][:])

    @classmethod
    def open_new(cls, fname, natom, box, crds=True, vels=False, frcs=False, title="", protocolWork=False, alchemicalLambda=False):
        """Same arguments and resulting layout as the reference's NetCDF4Traj.open_new (formats.py:504-650; REMD variables,
        which BLUES never requests, are not offered)."""
        self = cls(fname, "w")
        nc = self._nc
        self.hasbox, self.hascrds, self.hasvels, self.hasfrcs = bool(box), bool(crds), bool(vels), bool(frcs)
        self.hasprotocolWork, self.hasalchemicalLambda = bool(protocolWork), bool(alchemicalLambda)
        nc.Conventions = "AMBER"; nc.ConventionVersion = "1.0"
        nc.application = "blues_amd"; nc.program = "blues_amd"; nc.programVersion = "1"
        nc.title = title or "blues_amd trajectory"
        nc.createDimension("frame", None); nc.createDimension("spatial", 3); nc.createDimension("atom", int(natom))
        self.atom = int(natom)
        v = nc.createVariable("spatial", "c", ("spatial",)); v[:] = np.array(list("xyz"), dtype="S1")
        if self.hasbox:
            nc.createDimension("cell_spatial", 3); nc.createDimension("cell_angular", 3); nc.createDimension("label", 5)
            v = nc.createVariable("cell_spatial", "c", ("cell_spatial",)); v[:] = np.array(list("abc"), dtype="S1")
            v = nc.createVariable("cell_angular", "c", ("cell_angular", "label"))
            v[:] = np.array([list("alpha"), list("beta "), list("gamma")], dtype="S1")
        nc.createVariable("time", "f", ("frame",)).units = "picosecond"
        if self.hascrds:
            nc.createVariable("coordinates", "f", ("frame", "atom", "spatial")).units = "angstrom"
        if self.hasvels:
            v = nc.createVariable("velocities", "f", ("frame", "atom", "spatial"))
            v.units = "angstrom/picosecond"; v.scale_factor = AMBER_VELOCITY_SCALE
        if self.hasfrcs:
            nc.createVariable("forces", "f", ("frame", "atom", "spatial")).units = "kilocalorie/mole/angstrom"
        if self.hasbox:
            nc.createVariable("cell_lengths", "d", ("frame", "cell_spatial")).units = "angstrom"
            nc.createVariable("cell_angles", "d", ("frame", "cell_angular")).units = "degree"
        if self.hasprotocolWork:
            nc.createVariable("protocolWork", "f", ("frame",)).units = "kT"
        if self.hasalchemicalLambda:
            nc.createVariable("alchemicalLambda", "f", ("frame",)).units = "unitless"
        return self

    # ---- writing: one call per frame and variable; values arrive in OpenMM units (nm, nm/ps, kJ/mol/nm, ps)
    def _append(self, name, value):
        k = self._frames.get(name, 0)
        self._nc.variables[name][k] = value
        self._frames[name] = k + 1

    def add_coordinates(self, xyz_nm):
        self._append("coordinates", np.asarray(xyz_nm, dtype=np.float64).reshape(self.atom, 3) * _NM_TO_A)

    def add_velocities(self, v_nm_per_ps):
        self._append("velocities", np.asarray(v_nm_per_ps, dtype=np.float64).reshape(self.atom, 3) * (_NM_TO_A / AMBER_VELOCITY_SCALE))

    def add_forces(self, f_kj_per_mol_nm):
        self._append("forces", np.asarray(f_kj_per_mol_nm, dtype=np.float64).reshape(self.atom, 3) * _KJ_PER_NM_TO_KCAL_PER_A)

    def add_cell_lengths_angles(self, lengths_nm, angles_deg=(90.0, 90.0, 90.0)):
        self._append("cell_lengths", np.asarray(lengths_nm, dtype=np.float64) * _NM_TO_A)
        self._append("cell_angles", np.asarray(angles_deg, dtype=np.float64))

    def add_time(self, t_ps):
        self._append("time", float(t_ps))

    def add_protocolWork(self, work_kT):
        self._append("protocolWork", float(work_kT)); self.flush()

    def add_alchemicalLambda(self, lam):
        self._append("alchemicalLambda", float(lam)); self.flush()

    def flush(self):
        self._nc.flush()

    def close(self):
        if self._nc is not None:
            self._nc.close()
            self._nc = None

    # ---- reading (OpenMM units back)
    def _var(self, name):
        return np.array(self._nc.variables[name][:])

    @property
    def coordinates(self): return self._var("coordinates").astype(np.float64) / _NM_TO_A
    @property
    def velocities(self): return self._var("velocities").astype(np.float64) * (AMBER_VELOCITY_SCALE / _NM_TO_A)
    @property
    def forces(self): return self._var("forces").astype(np.float64) / _KJ_PER_NM_TO_KCAL_PER_A
    @property
    def cell_lengths(self): return self._var("cell_lengths") / _NM_TO_A
    @property
    def time(self): return self._var("time")
    @property
    def protocolWork(self): return self._var("protocolWork")
    @property
    def alchemicalLambda(self): return self._var("alchemicalLambda")


class AmberNetCDFRestart(object):
    """AMBER NetCDF restart file (Conventions "AMBERRESTART", version 1.0): what the reference writes through
    parmed.openmm.reporters.RestartReporter(outfname + '.rst7', netcdf=True) (reference blues/reporters.py:217-225) and loads back
    through parmed.amber.Rst7 (reference blues/settings.py:76-85).  One frame, no `frame` dimension, everything in double:
    coordinates [atom][spatial] (angstrom), velocities [atom][spatial] (angstrom/picosecond with scale_factor 20.455), time (ps),
    cell_lengths / cell_angles.  NetCDF-3 with 64-bit offsets through scipy, as the trajectory above."""

    @staticmethod
    def is_netcdf(path):
        with open(path, "rb") as fh:
            return fh.read(3) == b"CDF"

    @staticmethod
    def write(fname, positions_nm, velocities_nm_ps=None, box_nm=None, time_ps=0.0, title="", angles_deg=(90.0, 90.0, 90.0)):
        from scipy.io import netcdf_file
        x = np.asarray(positions_nm, dtype=np.float64).reshape(-1, 3)
        nc = netcdf_file(fname, "w", version=2, mmap=False)
        try:
            nc.Conventions = "AMBERRESTART"; nc.ConventionVersion = "1.0"
            nc.application = "blues_amd"; nc.program = "blues_amd"; nc.programVersion = "1"
            nc.title = title or "blues_amd restart"
            nc.createDimension("spatial", 3); nc.createDimension("atom", len(x))
            v = nc.createVariable("spatial", "c", ("spatial",)); v[:] = np.array(list("xyz"), dtype="S1")
            t = nc.createVariable("time", "d", ()); t.units = "picosecond"; t[...] = float(time_ps)   # (a scalar variable; scipy's assignValue indexes it with [:])
            c = nc.createVariable("coordinates", "d", ("atom", "spatial")); c.units = "angstrom"; c[:] = x * _NM_TO_A
            if velocities_nm_ps is not None:
                w = nc.createVariable("velocities", "d", ("atom", "spatial")); w.units = "angstrom/picosecond"; w.scale_factor = np.float64(AMBER_VELOCITY_SCALE)   # (a double, as the restart convention has it)
                w[:] = np.asarray(velocities_nm_ps, dtype=np.float64).reshape(-1, 3) * (_NM_TO_A / AMBER_VELOCITY_SCALE)
            if box_nm is not None:
                nc.createDimension("cell_spatial", 3); nc.createDimension("cell_angular", 3); nc.createDimension("label", 5)
                v = nc.createVariable("cell_spatial", "c", ("cell_spatial",)); v[:] = np.array(list("abc"), dtype="S1")
                v = nc.createVariable("cell_angular", "c", ("cell_angular", "label")); v[:] = np.array([list("alpha"), list("beta "), list("gamma")], dtype="S1")
                l = nc.createVariable("cell_lengths", "d", ("cell_spatial",)); l.units = "angstrom"; l[:] = np.asarray(box_nm, dtype=np.float64).reshape(-1)[:3] * _NM_TO_A
                g = nc.createVariable("cell_angles", "d", ("cell_angular",)); g.units = "degree"; g[:] = np.asarray(angles_deg, dtype=np.float64)
        finally:
            nc.close()

    @staticmethod
    def read(fname):
        """-> (positions nm, velocities nm/ps or None, box nm or None, time ps)"""
        from scipy.io import netcdf_file
        nc = netcdf_file(fname, "r", mmap=False)
        try:
            conv = getattr(nc, "Conventions", b"")
            conv = conv.decode() if isinstance(conv, bytes) else str(conv)
            if "AMBERRESTART" not in conv:
                raise ValueError("%s: NetCDF file with Conventions %r, not an AMBERRESTART file" % (fname, conv))
            v = nc.variables
            pos = np.array(v["coordinates"][:], dtype=np.float64) / _NM_TO_A
            vel = None
            if "velocities" in v:
                scale = float(getattr(v["velocities"], "scale_factor", 1.0))
                vel = np.array(v["velocities"][:], dtype=np.float64) * (scale / _NM_TO_A)
            box = np.array(v["cell_lengths"][:], dtype=np.float64) / _NM_TO_A if "cell_lengths" in v else None
            time = float(np.array(v["time"].getValue())) if "time" in v else 0.0
            return pos, vel, box, time
        finally:
            nc.close()
