"""The OpenMM-like object surface BLUES consumes from `simulations.ncmc`
(SURVEY.md section 8b): Simulation / Context / State, backed by the C-ABI engine.

Method names, argument meaning and error behaviour follow the OpenMM calls the
reference makes (file:line given per method); quantities are `blues_amd.unit.Quantity`.
"""
import numpy as np

from . import unit
from .engine import EngineError, NativeEngine


class PeriodicWrapper(object):
    """enforcePeriodicBox=True of Context.getState (reference blues/simulation.py:874-881): positions are reported with the centre of
    every MOLECULE inside the box, molecules kept whole, as OpenMM does.  Molecules are the connected components of the bond and
    constraint graph.  The shift of an atom is a lattice vector of its molecule, found from that molecule's atoms alone, so the few
    atoms a Move reads (positions[atom_indices]) are wrapped without fetching the rest."""

    def __init__(self, system):
        # the molecule table depends on the bond / constraint graph alone: R contexts built from one System object share it
        # (26 ms of union-find per context otherwise -- a third of a rank's set-up time at 512 chains)
        # (keyed on the graph itself, not on the atom count: a copy.copy() of the System shares this attribute, and its bonds or
        # constraints may have been edited since)
        graph = (system.n_atoms,) + tuple(hash(np.ascontiguousarray(getattr(system, name, np.zeros((0, 2))), dtype=np.int64).tobytes()) for name in ("bond_atoms", "constraint_atoms"))
        cached = getattr(system, "_periodic_molecules", None)
        if cached is not None and cached[0] == graph:
            _, self.molecule_of, self._order, self._start = cached
            self.box = np.asarray(system.box, dtype=np.float64).reshape(-1)[:3].copy() if np.size(system.box) == 3 else np.diag(np.asarray(system.box, dtype=np.float64).reshape(3, 3)).copy()
            return
        n = system.n_atoms
        parent = list(range(n))

        def find(i):
            while parent[i] != i:
                parent[i] = parent[parent[i]]
                i = parent[i]
            return i
        for pairs in (getattr(system, "bond_atoms", []), getattr(system, "constraint_atoms", [])):
            for a, b in np.asarray(pairs, dtype=np.int64).reshape(-1, 2):
                ra, rb = find(int(a)), find(int(b))
                if ra != rb:
                    parent[max(ra, rb)] = min(ra, rb)
        roots = np.array([find(i) for i in range(n)], dtype=np.int64)
        _, self.molecule_of = np.unique(roots, return_inverse=True)
        order = np.argsort(self.molecule_of, kind="stable")
        self._order = order
        self._start = np.searchsorted(self.molecule_of[order], np.arange(self.molecule_of.max() + 2))
        try:
            system._periodic_molecules = (graph, self.molecule_of, self._order, self._start)
        except Exception:
            pass   # (a System that takes no attributes: every context computes its own)
        self.box = np.asarray(system.box, dtype=np.float64).reshape(-1)[:3].copy() if np.size(system.box) == 3 else np.diag(np.asarray(system.box, dtype=np.float64).reshape(3, 3)).copy()

    def atoms_of(self, molecule):
        return self._order[self._start[molecule]:self._start[molecule + 1]]

    def wrap_all(self, x, box=None):
        box = self.box if box is None else box
        x = np.asarray(x, dtype=np.float64)
        counts = np.diff(self._start).astype(np.float64)
        centre = np.zeros((len(counts), 3))
        np.add.at(centre, self.molecule_of, x)
        centre /= counts[:, None]
        return x - (np.floor(centre / box) * box)[self.molecule_of]

    def shifts_for(self, indices, read_atoms, box=None):
        """Lattice shifts of the atoms `indices`; read_atoms(idx) -> coordinates of any atoms (used for their molecules' other atoms)."""
        box = self.box if box is None else box
        indices = np.asarray(indices, dtype=np.int64)
        mols = np.unique(self.molecule_of[indices])
        need = np.concatenate([self.atoms_of(m) for m in mols])
        xyz = np.asarray(read_atoms(need), dtype=np.float64)
        shift_of = {}
        pos = 0
        for m in mols:
            k = len(self.atoms_of(m))
            shift_of[m] = -np.floor(xyz[pos:pos + k].mean(0) / box) * box
            pos += k
        return np.array([shift_of[m] for m in self.molecule_of[indices]])


class State(object):
    """Result of Context.getState (reference blues/simulation.py:904-910)."""

    def __init__(self, positions=None, velocities=None, forces=None, potential=None, kinetic=None, box=None, time=0.0,
                 parameters=None, snapshot=None, wrapper=None):
        self._x, self._v, self._f, self._pe, self._ke, self._box, self._t, self._par = positions, velocities, forces, potential, kinetic, box, time, parameters
        self._snap = snapshot   # positions / velocities still on the device (engine.DeviceSnapshot); downloaded on demand
        self._wrapper = wrapper  # enforcePeriodicBox: applied to whatever part of the positions reaches the host

    @staticmethod
    def _need(v, what):
        if v is None:
            raise Exception("Invoked %s on a State which does not contain it." % what)
        return v

    def getPositions(self, asNumpy=False):
        if self._snap is not None and (self._snap.what & 1):
            if asNumpy:
                return unit.DeviceQuantity(self._snap, 1, "nanometer", wrapper=self._wrapper)
            x = self._snap.read(1)
            if self._wrapper is not None:
                x = self._wrapper.wrap_all(x)
            return unit.Quantity([tuple(r) for r in x], "nanometer")
        x = self._need(self._x, "getPositions()")
        if self._wrapper is not None:
            x = self._wrapper.wrap_all(x)
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
        self._wrapper = None   # built on first use (enforcePeriodicBox=True)

    def periodic_wrapper(self):
        if self._wrapper is None:
            self._wrapper = PeriodicWrapper(self._system)
        # (asked for by every state capture of every chain: the box is re-read only when the engine's copy of it was replaced --
        # NativeEngine.set_box drops that copy)
        ref = self._engine.__dict__.get("_box_copy")
        if ref is None or ref is not self.__dict__.get("_wrapper_box_ref"):
            box = np.asarray(self._engine.get_box(), dtype=np.float64)
            self._wrapper.box = np.diag(box).copy() if box.shape == (3, 3) else box.reshape(-1)[:3].copy()
            self._wrapper_box_ref = self._engine.__dict__.get("_box_copy")
        return self._wrapper

    def box_vector_quantities(self):
        """The three box vectors as Quantities (a State's 'box_vectors'); the same objects until the box changes (read-only by convention)."""
        ref = self._engine.__dict__.get("_box_copy")
        hit = self.__dict__.get("_box_q")
        if hit is None or ref is None or hit[0] is not ref:
            rows = self._engine.get_box()
            hit = (self._engine.__dict__.get("_box_copy"), [unit.Quantity(np.array(row), "nanometer") for row in rows])
            self._box_q = hit
        return hit[1]

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
        # (only where the potential itself is periodic: wrapped coordinates handed to a NoCutoff system -- the ethylene fixture of
        # reference blues/tests/test_ethylene.py -- would change its energies)
        periodic = getattr(self._system, "nonbonded_method", 1) != 0
        wrapper = self.periodic_wrapper() if (enforcePeriodicBox and getPositions and periodic and hasattr(self._system, "n_atoms")) else None
        return State(x, v, f, pe, ke, e.get_box(), self._time, par, snapshot=snap, wrapper=wrapper)

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

    def minimizeEnergy(self, tolerance=10.0, maxIterations=0):
        """app.Simulation.minimizeEnergy (the reference calls it in its test fixtures only, blues/tests/test_simulation.py:139-141).
        OpenMM runs L-BFGS; here: steepest descent on the engine's forces with an adaptive step -- accepted while the potential
        energy falls, halved otherwise -- and the constraints re-imposed after every displacement (the procedure of the oracle's
        orc_minimize).  tolerance: largest force component (kJ/mol/nm) at which to stop; maxIterations 0: up to 500."""
        e = self.context._engine
        s = self.system
        tol = unit.value_in(tolerance, "kilojoule/(nanometer*mole)") if isinstance(tolerance, unit.Quantity) else float(tolerance)
        mobile = np.asarray(s.mass) > 0
        ca = np.asarray(getattr(s, "constraint_atoms", []), dtype=np.int64).reshape(-1, 2)
        cd = np.asarray(getattr(s, "constraint_dist", []), dtype=np.float64).reshape(-1)
        live = mobile[ca[:, 0]] | mobile[ca[:, 1]] if len(ca) else np.zeros(0, bool)
        ca, cd = ca[live], cd[live]
        w = np.where(mobile, 1.0 / np.where(mobile, s.mass, 1.0), 0.0)
        box = np.diag(np.asarray(e.get_box()).reshape(3, 3))

        def constrain(x, ref):   # SHAKE along the reference bond directions, all constraints per sweep
            for _ in range(200):
                d = x[ca[:, 0]] - x[ca[:, 1]]; d -= box * np.round(d / box)
                diff = (d * d).sum(1) - cd * cd
                if not len(ca) or np.abs(diff).max() <= 2e-8 * (cd * cd).max():
                    break
                r = ref[ca[:, 0]] - ref[ca[:, 1]]; r -= box * np.round(r / box)
                g = diff / (2.0 * (w[ca[:, 0]] + w[ca[:, 1]]) * (d * r).sum(1))
                np.add.at(x, ca[:, 0], -(g * w[ca[:, 0]])[:, None] * r)
                np.add.at(x, ca[:, 1], (g * w[ca[:, 1]])[:, None] * r)
            return x
        x = e.get_positions()
        x = constrain(x.copy(), x)
        e.set_positions(x)
        E, f = e.potential_energy(), e.get_forces()
        step = 0.01
        for _ in range(int(maxIterations) if maxIterations else 500):
            fmax = np.abs(f[mobile]).max() if mobile.any() else 0.0
            if fmax <= tol or step < 1e-7:
                break
            xn = x.copy(); xn[mobile] += step * f[mobile] / fmax
            xn = constrain(xn, x)
            e.set_positions(xn)
            En = e.potential_energy()
            if En < E:
                x, E, f, step = xn, En, e.get_forces(), step * 1.2
            else:
                e.set_positions(x); step *= 0.5
        v = e.get_velocities()
        if np.any(v):
            e.set_velocities(v)   # (unchanged; keeps the engine's velocity bookkeeping in step with the new positions)

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
        if self.barostat is not None and self.barostat.frequency > 0 and chunk > 0:   # (frequency 0: OpenMM's "barostat disabled")
            # OpenMM's updateContextState runs at the top of every step: it increments its counter and makes the attempt when the
            # counter reaches `frequency` -- i.e. BEFORE the frequency-th step, after frequency - 1 completed ones.  The attempt is
            # made here, by whoever plans the next chunk: Simulation.step for a lone chain, BatchedBLUESSimulation._advance for the
            # chains of a replica batch (each member has its own box: the batch's records carry it per member)
            if self._barostat_count + 1 >= self.barostat.frequency:
                self.barostat.attempt(self.context._engine, self.system)
                self._barostat_count = -1      # (the step that follows is the first of the next period)
                left = 1
            else:
                left = self.barostat.frequency - 1 - self._barostat_count
            if left < chunk:
                chunk = left
                due = [(r, nxt) for r, nxt in due if nxt[0] == chunk]
        return chunk, due

    def _commit_chunk(self, chunk, due):
        self.currentStep += chunk
        if self.barostat is not None and self.barostat.frequency > 0:
            self._barostat_count += chunk
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
            chunk, due = self._plan_chunk(end)     # (makes the barostat's attempt when one is due)
            self.integrator.step(chunk)
            self._commit_chunk(chunk, due)
