"""Reporters of the NCMC / MD simulations with the OpenMM reporter protocol (describeNextReport / report), after
reference blues/reporters.py: NetCDF4Reporter (:731-865) and BLUESStateDataReporter (:436-729).  Both can be told to
fire at given frame indices instead of a fixed interval (the reference uses that for "the move step" and "the last
frame" of a switch); Simulation.step cuts its chunks there (blues_amd/context.py)."""
import sys
import time as _time

import numpy as np

from . import unit
from .formats import AmberNetCDFTraj

_R_KJ = 0.0083144626   # MOLAR_GAS_CONSTANT_R in kJ/mol/K


class _Schedule(object):
    """reportInterval or frame_indices -> steps until the next report (reference reporters.py:777-802, 534-561)."""

    def __init__(self, reportInterval, frame_indices):
        self._reportInterval = int(reportInterval)
        # "If simulation.currentStep = 1, store the frame from the previous step": indices are shifted down by one
        self.frame_indices = [int(x) - 1 for x in frame_indices] if frame_indices else []

    def steps_to_next(self, simulation):
        if self.frame_indices:
            return 1 if simulation.currentStep in self.frame_indices else -1
        return self._reportInterval - simulation.currentStep % self._reportInterval


class NetCDF4Reporter(_Schedule):
    """AMBER NetCDF trajectory of a simulation, optionally with protocolWork [kT] and alchemicalLambda per frame."""

    def __init__(self, file, reportInterval=1, frame_indices=[], crds=True, vels=False, frcs=False, protocolWork=False, alchemicalLambda=False):
        _Schedule.__init__(self, reportInterval, frame_indices)
        self.fname = file
        self.crds, self.vels, self.frcs, self.protocolWork, self.alchemicalLambda = crds, vels, frcs, protocolWork, alchemicalLambda
        self._out = None

    def describeNextReport(self, simulation):
        return (self.steps_to_next(simulation), self.crds, self.vels, self.frcs, False)

    def report(self, simulation, state):
        x = unit.value_in(state.getPositions(asNumpy=True), "nanometer") if self.crds else None
        v = unit.value_in(state.getVelocities(asNumpy=True), "nanometer/picosecond") if self.vels else None
        f = unit.value_in(state.getForces(asNumpy=True), "kilojoule/(nanometer*mole)") if self.frcs else None
        if self._out is None:   # first frame: lay the file out
            natom = len(x if x is not None else (v if v is not None else f))
            self.uses_pbc = True   # this engine is periodic by construction (reference: topology.getUnitCellDimensions() is not None)
            self._out = AmberNetCDFTraj.open_new(self.fname, natom, self.uses_pbc, self.crds, self.vels, self.frcs,
                                                 title="blues_amd trajectory", protocolWork=self.protocolWork, alchemicalLambda=self.alchemicalLambda)
        if self.uses_pbc:
            box = np.asarray(unit.value_in(state.getPeriodicBoxVectors(asNumpy=True), "nanometer")).reshape(3, 3)
            self._out.add_cell_lengths_angles(np.diag(box))
        if self.crds: self._out.add_coordinates(x)
        if self.vels: self._out.add_velocities(v)
        if self.frcs: self._out.add_forces(f)
        if self.protocolWork: self._out.add_protocolWork(simulation.integrator.get_protocol_work(dimensionless=True))
        if self.alchemicalLambda: self._out.add_alchemicalLambda(simulation.integrator.getGlobalVariableByName("lambda"))
        self._out.add_time(unit.value_in(state.getTime(), "picosecond"))
        self._out.flush()

    def close(self):
        if self._out is not None:
            self._out.close()
            self._out = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BLUESStateDataReporter(_Schedule):
    """One line of state data per report: '<title>: v1<sep>v2...', header '#"a"<sep>"b"...' once
    (column order as reference reporters.py:602-729)."""

    def __init__(self, file, reportInterval=1, frame_indices=[], title="", step=False, time=False, potentialEnergy=False,
                 kineticEnergy=False, totalEnergy=False, temperature=False, volume=False, density=False, progress=False,
                 remainingTime=False, speed=False, elapsedTime=False, separator="\t", systemMass=None, totalSteps=None,
                 protocolWork=False, alchemicalLambda=False, currentIter=False):
        _Schedule.__init__(self, reportInterval, frame_indices)
        if (progress or remainingTime) and totalSteps is None:
            raise ValueError("Reporting progress or remaining time requires total steps to be specified")
        self._own = isinstance(file, str)
        self._out = open(file, "w") if self._own else file
        self.title, self._separator = title, separator
        self._step, self._time, self._pe, self._ke, self._te, self._temp = step, time, potentialEnergy, kineticEnergy, totalEnergy, temperature
        self._volume, self._density, self._progress, self._remaining, self._speed, self._elapsed = volume, density, progress, remainingTime, speed, elapsedTime
        self._totalSteps, self._totalMass = totalSteps, systemMass
        self._protocolWork, self._alchemicalLambda, self._currentIter = protocolWork, alchemicalLambda, currentIter
        self._needEnergy = potentialEnergy or kineticEnergy or totalEnergy or temperature
        self._initialized = False

    def describeNextReport(self, simulation):
        return (self.steps_to_next(simulation), False, False, False, self._needEnergy)

    def _emit(self, text):
        if hasattr(self._out, "report"):
            self._out.report(text)
        elif hasattr(self._out, "info") and not hasattr(self._out, "write"):
            self._out.info(text)
        else:
            self._out.write(text + "\n")
            try:
                self._out.flush()
            except AttributeError:
                pass

    def _headers(self):
        h = []
        if self._currentIter: h.append("Iter")
        if self._progress: h.append("Progress (%)")
        if self._step: h.append("Step")
        if self._time: h.append("Time (ps)")
        if self._alchemicalLambda: h.append("alchemicalLambda")
        if self._protocolWork: h.append("protocolWork")
        if self._pe: h.append("Potential Energy (kJ/mole)")
        if self._ke: h.append("Kinetic Energy (kJ/mole)")
        if self._te: h.append("Total Energy (kJ/mole)")
        if self._temp: h.append("Temperature (K)")
        if self._volume: h.append("Box Volume (nm^3)")
        if self._density: h.append("Density (g/mL)")
        if self._speed: h.append("Speed (ns/day)")
        if self._elapsed: h.append("Elapsed Time (s)")
        if self._remaining: h.append("Time Remaining")
        return h

    def _init(self, simulation, state):
        system = simulation.context.getSystem()
        mass = np.asarray(system.mass, dtype=np.float64)
        # degrees of freedom as OpenMM's StateDataReporter counts them: 3 per massive particle, minus constraints, minus 3
        # when a CMMotionRemover is present
        ncons = sum(1 for (i, j) in np.asarray(system.constraint_atoms).reshape(-1, 2) if mass[i] > 0 or mass[j] > 0)
        self._dof = 3 * int((mass > 0).sum()) - ncons - (3 if getattr(system, "remove_cm_motion", False) else 0)
        if self._totalMass is None:
            self._totalMass = float(mass.sum())
        self._t0, self._time0, self._step0 = _time.time(), unit.value_in(state.getTime(), "picosecond"), simulation.currentStep
        self._emit('#"%s"' % ('"' + self._separator + '"').join(self._headers()))
        self._initialized = True

    def report(self, simulation, state):
        if not self._initialized:
            self._init(simulation, state)
        if self._needEnergy:
            e = state.getPotentialEnergy()._value
            if e != e:
                raise ValueError("Energy is NaN")
        vals = []
        box = np.asarray(unit.value_in(state.getPeriodicBoxVectors(asNumpy=True), "nanometer")).reshape(3, 3)
        volume = box[0, 0] * box[1, 1] * box[2, 2]
        now = _time.time()
        if self._currentIter: vals.append(getattr(simulation, "currentIter", 0))
        if self._progress: vals.append("%.1f%%" % (100.0 * simulation.currentStep / self._totalSteps))
        if self._step: vals.append(simulation.currentStep)
        if self._time: vals.append(unit.value_in(state.getTime(), "picosecond"))
        if self._alchemicalLambda: vals.append(simulation.integrator.getGlobalVariableByName("lambda"))
        if self._protocolWork: vals.append(simulation.integrator.get_protocol_work(dimensionless=True))
        if self._pe: vals.append(state.getPotentialEnergy()._value)
        if self._ke: vals.append(state.getKineticEnergy()._value)
        if self._te: vals.append(state.getKineticEnergy()._value + state.getPotentialEnergy()._value)
        if self._temp: vals.append(2.0 * state.getKineticEnergy()._value / (self._dof * _R_KJ))
        if self._volume: vals.append(volume)
        if self._density: vals.append(self._totalMass / volume * 1.66053906660e-3)   # amu/nm^3 -> g/mL
        if self._speed:
            dt_days = (now - self._t0) / 86400.0
            dns = (unit.value_in(state.getTime(), "picosecond") - self._time0) * 1e-3
            vals.append("%.3g" % (dns / dt_days) if dt_days > 0 else "--")
        if self._elapsed: vals.append(now - self._t0)
        if self._remaining:
            done = simulation.currentStep - self._step0
            vals.append("--" if done <= 0 else "%d s" % int((now - self._t0) * (self._totalSteps - simulation.currentStep) / done))
        self._emit("%s: %s" % (self.title, self._separator.join(str(v) for v in vals)))

    def close(self):
        if self._own and self._out is not None:
            self._out.close()
            self._out = None


class RestartReporter(_Schedule):
    """parmed.openmm.reporters.RestartReporter as the reference configures it (reference blues/reporters.py:217-225:
    RestartReporter(outfname + '.rst7', netcdf=True, **cfg)): every reportInterval steps the current positions, velocities and
    box are written as an Amber restart, overwriting the file unless write_multiple (then `<file>.<step>`)."""

    def __init__(self, file, reportInterval=1, write_multiple=False, netcdf=False, write_velocities=True, frame_indices=[]):
        super(RestartReporter, self).__init__(reportInterval, frame_indices)
        self.fname, self.write_multiple, self.netcdf, self.write_velocities = file, bool(write_multiple), bool(netcdf), bool(write_velocities)

    def describeNextReport(self, simulation):
        return (self.steps_to_next(simulation), True, self.write_velocities, False, False)

    def report(self, simulation, state):
        from .amber import write_rst7
        from . import unit
        fname = "%s.%d" % (self.fname, simulation.currentStep) if self.write_multiple else self.fname
        x = unit.value_in(state.getPositions(asNumpy=True), "nanometer")
        v = unit.value_in(state.getVelocities(asNumpy=True), "nanometer/picosecond") if self.write_velocities else None
        box = np.diag(np.asarray(unit.value_in(state.getPeriodicBoxVectors(asNumpy=True), "nanometer")).reshape(3, 3))
        write_rst7(fname, x, v, box, time_ps=unit.value_in(state.getTime(), "picosecond"), netcdf=self.netcdf)
