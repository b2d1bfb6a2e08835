"""The OpenMM-like object surface BLUES consumes from `simulations.ncmc`
(SURVEY.md section 8b): Simulation / Context / State, backed by the C-ABI engine.

Method names, argument meaning and error behaviour follow the OpenMM calls the
reference makes (file:line given per method); quantities are `blues_amd.unit.Quantity`.
"""
import numpy as np

from . import unit
from .engine import EngineError, NativeEngine


class State(object):
    """Result of Context.getState (reference blues/simulation.py:904-910)."""

    def __init__(self, positions=None, velocities=None, forces=None, potential=None, kinetic=None, box=None, time=0.0,
                 parameters=None, snapshot=None):
        self._x, self._v, self._f, self._pe, self._ke, self._box, self._t, self._par = positions, velocities, forces, potential, kinetic, box, time, parameters
        self._snap = snapshot   # positions / velocities still on the device (engine.DeviceSnapshot); downloaded on demand

    @staticmethod
    def _need(v, what):
        if v is None:
            raise Exception("Invoked %s on a State which does not contain it." % what)
        return v

    def getPositions(self, asNumpy=False):
        if self._snap is not None and (self._snap.what & 1):
            if asNumpy:
                return unit.DeviceQuantity(self._snap, 1, "nanometer")
            return unit.Quantity([tuple(r) for r in self._snap.read(1)], "nanometer")
        x = self._need(self._x, "getPositions()")
        return unit.Quantity(x if asNumpy else [tuple(r) for r in x], "nanometer") if not asNumpy else unit.Quantity(x, "nanometer")

    def getVelocities(self, asNumpy=False):
        if self._snap is not None and (self._snap.what & 2):
            return unit.DeviceQuantity(self._snap, 2, "nanometer/picosecond")
        return unit.Quantity(self._need(self._v, "getVelocities()"), "nanometer/picosecond")

    def getForces(self, asNumpy=False):
        return unit.Quantity(self._need(self._f, "getForces()"), "kilojoule/(nanometer*mole)")

    def getPotentialEnergy(self):
        return unit.Quantity(self._need(self._pe, "getPotentialEnergy()"), "kilojoule/mole")

    def getKineticEnergy(self):
        return unit.Quantity(self._need(self._ke, "getKineticEnergy()"), "kilojoule/mole")

    def getPeriodicBoxVectors(self, asNumpy=False):
        if asNumpy:
            return unit.Quantity(np.array(self._box), "nanometer")
        return [unit.Quantity(np.array(row), "nanometer") for row in self._box]

    def getPeriodicBoxVolume(self):
        return float(np.linalg.det(self._box))

    def getTime(self):
        return unit.Quantity(self._t, "picosecond")

    def getParameters(self):
        return dict(self._need(self._par, "getParameters()"))


class Platform(object):
    def getName(self):
        return "HIP-gfx950"

    def getSpeed(self):
        return 100.0

    def getPropertyNames(self):
        return ["DeviceIndex", "Precision"]


class Context(object):
    """openmm.Context stand-in bound to one engine handle (one replica on one GPU)."""

    def __init__(self, system, integrator, device=0, precision="mixed", replica=0):
        self._system = system
        self._integrator = integrator  # BLUES reads context._integrator (reference blues/simulation.py:1117,1130,1184)
        self._device = int(device)
        self._precision = precision
        prec = {"mixed": 0, "single": 0, "double": 1}[precision] if isinstance(precision, str) else int(precision)
        self._engine = NativeEngine(system, integrator.to_data(replica=replica, precision=prec), device=device)
        integrator._bind(self._engine)
        self._time = 0.0
        self._platform = Platform()

    def getState(self, getPositions=False, getVelocities=False, getForces=False, getEnergy=False, getParameters=False,
                 enforcePeriodicBox=False, groups=-1):
        """context.getState(...) (reference blues/simulation.py:905, blues/moves.py:292, positional form moves.py:1218)."""
        e = self._engine
        x = v = snap = None
        pe = ke = None
        if getEnergy:   # before the snapshot, so that the snapshot carries the energy of its positions
            pe, ke = e.energies() if hasattr(e, "energies") else (e.potential_energy(), e.kinetic_energy())
        if (getPositions or getVelocities) and hasattr(e, "snapshot"):
            snap = e.snapshot(positions=bool(getPositions), velocities=bool(getVelocities))   # stays in HBM until somebody reads it
        else:
            x = e.get_positions() if getPositions else None
            v = e.get_velocities() if getVelocities else None
        f = e.get_forces() if getForces else None
        par = None
        if getParameters:
            par = {"lambda_sterics": e.get_global("lambda_sterics"), "lambda_electrostatics": e.get_global("lambda_electrostatics")}
        return State(x, v, f, pe, ke, e.get_box(), self._time, par, snapshot=snap)

    def setPositions(self, positions):
        """reference blues/simulation.py:960, blues/moves.py:307"""
        snap = positions.on_device() if isinstance(positions, unit.DeviceQuantity) else None
        if snap is not None and hasattr(self._engine, "set_positions_from_snapshot"):
            idx, rows = positions.pending_edits()
            if len(idx) == 0:
                self._engine.set_positions_from_snapshot(snap)   # a State handed back unchanged: device-to-device
                return
            f = 1.0 / unit.Quantity(1.0, "nanometer").value_in_unit(positions.unit)   # the Quantity's unit -> nm
            if self._engine.set_positions_from_snapshot_edited(snap, idx, rows * f):  # a few atoms edited by a Move
                return
        self._engine.set_positions(unit.value_in(positions, "nanometer"))

    def setVelocities(self, velocities):
        """reference blues/simulation.py:962"""
        snap = velocities.on_device() if isinstance(velocities, unit.DeviceQuantity) else None
        if snap is not None and hasattr(self._engine, "set_velocities_from_snapshot"):
            self._engine.set_velocities_from_snapshot(snap)
        else:
            self._engine.set_velocities(unit.value_in(velocities, "nanometer/picosecond"))

    def setPeriodicBoxVectors(self, a, b, c):
        """reference blues/simulation.py:958"""
        box = np.array([unit.value_in(a, "nanometer"), unit.value_in(b, "nanometer"), unit.value_in(c, "nanometer")], dtype=np.float64)
        key = box.tobytes()
        if key == getattr(self._engine, "_box_key", None):
            return   # the vectors this context was given last (every hand-over of an NVT chain): nothing to do
        self._engine.set_box(box.reshape(3, 3).diagonal().copy() if np.allclose(box, np.diag(np.diag(box))) else box)
        self._engine._box_key = key

    def setVelocitiesToTemperature(self, temperature, randomSeed=None):
        """reference blues/simulation.py:743, 1187"""
        if randomSeed is None:
            randomSeed = np.random.randint(0, 2 ** 31 - 1)
        self._engine.set_velocities_to_temperature(unit.value_in(temperature, "kelvin"), randomSeed)

    def getParameter(self, name):
        return self._engine.get_global(name)

    def setParameter(self, name, value):
        self._engine.set_global(name, value)

    def getIntegrator(self):
        return self._integrator

    def getPlatform(self):
        return self._platform

    def getSystem(self):
        return self._system


class Simulation(object):
    """openmm.app.Simulation stand-in (reference blues/simulation.py:732-737): `context` is read AND
    assigned by BLUES (simulation.py:1037,1070,1079,1086); reporters follow the OpenMM protocol
    describeNextReport(sim) -> (steps, pos, vel, frc, ene) / report(sim, state)."""

    def __init__(self, topology, system, integrator, platform=None, platformProperties=None, device=0, precision="mixed", replica=0):
        props = dict(platformProperties or {})
        device = int(props.get("DeviceIndex", device))
        precision = props.get("Precision", precision)
        self.topology = topology
        self.system = system
        self.integrator = integrator
        self.context = Context(system, integrator, device=device, precision=precision, replica=replica)
        self.reporters = []
        self.currentStep = 0
        self.currentIter = 0
        # MonteCarloBarostat of the System (reference blues/simulation.py:603-626): applied before every `frequency`-th step
        self.barostat = None
        if getattr(system, "barostat", None):
            from .barostat import MonteCarloBarostat
            p_bar, temp, freq = system.barostat
            self.barostat = MonteCarloBarostat(p_bar, temp, freq, seed=getattr(integrator, "_seed", 0) + 7919 * (replica + 1))
            self._barostat_count = 0

    def minimizeEnergy(self, tolerance=None, maxIterations=0):
        raise NotImplementedError("energy minimisation is outside the NCMC switching path (tests only in the reference)")

    def _plan_chunk(self, end):
        """Steps that can be taken before the next reporter is due (at most up to `end`), and the reporters asked."""
        chunk = end - self.currentStep
        due = []
        for r in self.reporters:
            nxt = r.describeNextReport(self)
            due.append((r, nxt))
            if 0 < nxt[0] < chunk:
                chunk = nxt[0]
            # a reporter that fires at given frame indices (reference blues/reporters.py:791-796) only says "now" when the
            # step counter sits ON an index; the reference steps one at a time and asks before every step, so the fused
            # stepping here must stop at the next index to ask again
            ahead = [i - self.currentStep for i in (getattr(r, "frame_indices", None) or []) if i > self.currentStep]
            if ahead and min(ahead) < chunk:
                chunk = min(ahead)
        return chunk, due

    def _commit_chunk(self, chunk, due):
        self.currentStep += chunk
        self.context._time += chunk * self.integrator._timestep
        for r, nxt in due:
            if nxt[0] == chunk:
                st = self.context.getState(getPositions=bool(nxt[1]), getVelocities=bool(nxt[2]), getForces=bool(nxt[3]),
                                           getEnergy=bool(nxt[4]), getParameters=True)
                r.report(self, st)

    def step(self, steps):
        """app.Simulation.step (reference blues/simulation.py:1082): advance, honouring reporter intervals."""
        end = self.currentStep + int(steps)
        while self.currentStep < end:
            chunk, due = self._plan_chunk(end)
            if self.barostat is not None:
                # OpenMM counts steps in updateContextState and makes its attempt when the count reaches `frequency`, before that step
                if self._barostat_count >= self.barostat.frequency:
                    self.barostat.attempt(self.context._engine, self.system)
                    self._barostat_count = 0
                left = self.barostat.frequency - self._barostat_count
                if left < chunk:
                    chunk = left
                    due = [(r, nxt) for r, nxt in due if nxt[0] == chunk]
                self._barostat_count += chunk
            self.integrator.step(chunk)
            self._commit_chunk(chunk, due)
