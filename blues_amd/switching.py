"""Host-side mirrors of the switching integrators of reference blues/switching.py:
NCMCAlchemicalIntegrator (:804-1080), NCMCVVAlchemicalIntegrator (:1083-1241) and
NCMCGHMCAlchemicalIntegrator (:1244-1360).  The module is dead code in the reference
(nothing imports it); the classes keep its constructor signatures,
attributes (`direction`, `kT`, `nsteps`, `alchemical_functions`, `has_statistics`) and
accessors (`getTotalWork`, `getShadowWork`, `getProtocolWork` in kT, `getLogAcceptanceProbability`,
`getStatistics`, `get_step`, `reset`).  No arithmetic here: a context binds them to the C-ABI
engine (`switching_mode` of BluesIntegratorDesc, include/blues_engine.h).

Two places where the reference's GHMC program cannot be followed to the letter (documented in
oracle/blues_oracle.c `ghmc_step` / `switching_step`): it sums the new kinetic energy into an
undeclared variable and forms the acceptance from the old one (OpenMM would reject the program),
and it leaves `Epert` stale after the GHMC step of the first-step block.  The evident intent is used.
"""
import numpy as np

from . import lepton, unit
from ._abi import SWITCH_GHMC, SWITCH_VV, IntegratorData

KB = 0.0083144626  # kJ/mol/K


class NCMCAlchemicalIntegrator(object):
    """Helper base class (reference blues/switching.py:804-1080)."""

    _mode = None

    def __init__(self, temperature, system, functions, nsteps, steps_per_propagation, timestep, direction):
        if direction not in ['insert', 'delete', 'flux']:
            raise Exception("'direction' must be one of ['insert', 'delete', 'flux']; was '%s' instead" % direction)
        self.direction = direction
        self._temperature = unit.value_in(temperature, "kelvin")
        self.kT = unit.Quantity(KB * self._temperature, "kilojoule/mole")
        self.has_statistics = False
        self.nsteps = int(nsteps)
        self.alchemical_functions = dict(functions)
        # (the reference keeps the functions of parameters that exist in the System; the engine knows these two)
        self.system_parameters = {"lambda_sterics", "lambda_electrostatics"}
        unknown = set(self.alchemical_functions) - self.system_parameters
        self._ignored_functions = sorted(unknown)   # parameters no force of the System has are skipped, as in switching.py:917-919
        self._psteps = int(steps_per_propagation)
        self._timestep = unit.value_in(timestep, "picosecond")
        self._collision_rate = 0.0
        self._constraint_tolerance = 1e-8
        self._seed = 0
        self._engine = None
        for v in self.alchemical_functions.values():
            lepton.compile_expression(v)
        if self.nsteps < 0 or self._psteps < 1:
            raise ValueError("nsteps >= 0 and steps_per_propagation >= 1")

    # ---- the lambda values the program visits (switching.py:855-900): entry i of the tables handed to the engine
    def _lambda_path(self):
        n = max(1, self.nsteps)
        if self.direction == 'insert':
            return [i / float(n) for i in range(n + 1)]
        if self.direction == 'delete':
            return [(n - i) / float(n) for i in range(n + 1)]
        return [1.0] + [i / float(n) for i in range(1, n + 1)]   # 'flux': reset to 1.0, then (step+1)/nsteps

    def to_data(self, replica=0, precision=0):
        lam = self._lambda_path()
        tab = {}
        for name in ("lambda_sterics", "lambda_electrostatics"):
            expr = self.alchemical_functions.get(name)
            # a parameter without a function keeps its Context value; the alchemical System's default is 1.0
            tab[name] = np.array([lepton.compile_expression(expr)(**{"lambda": t}) if expr is not None else 1.0 for t in lam])
        return IntegratorData(timestep=self._timestep, temperature=self._temperature, nsteps_neq=max(1, self.nsteps),
                              lambda_sterics=tab["lambda_sterics"], lambda_electrostatics=tab["lambda_electrostatics"], splitting="",
                              collision_rate=self._collision_rate, nprop=1, prop_lambda_min=2.0, prop_lambda_max=-1.0,
                              constraint_tolerance=self._constraint_tolerance, seed=self._seed, replica=replica, precision=precision,
                              switching_mode=self._mode, steps_per_propagation=self._psteps if self.nsteps > 0 else 0)

    # ---- OpenMM Integrator surface
    def getTemperature(self):
        return unit.Quantity(self._temperature, "kelvin")

    def getStepSize(self):
        return unit.Quantity(self._timestep, "picosecond")

    def setRandomNumberSeed(self, seed):
        self._seed = int(seed)

    def getRandomNumberSeed(self):
        return self._seed

    def _bind(self, engine):
        self._engine = engine

    def _need(self):
        if self._engine is None:
            raise RuntimeError("integrator is not bound to a context")
        return self._engine

    def step(self, n):
        self._need().step(int(n))

    def getGlobalVariableByName(self, name):
        if self._engine is None:
            return {"nsteps": float(self.nsteps), "psteps": float(self._psteps), "kT": KB * self._temperature}.get(name, 0.0)
        kT = KB * self._temperature
        if name in ("protocol_work", "shadow_work", "total_work"):      # the reference keeps these in kT
            return self._engine.get_global(name) / kT
        if name == "initial_reduced_potential":
            return self._engine.get_global("initial_energy") / kT
        if name == "final_reduced_potential":
            return self._engine.get_global("final_energy") / kT
        if name == "psteps":
            return float(self._psteps)
        return self._engine.get_global(name)

    def setGlobalVariableByName(self, name, value):
        kT = KB * self._temperature
        if name in ("protocol_work", "shadow_work", "total_work"):
            value = float(value) * kT
        self._need().set_global(name, float(value))

    # ---- reference blues/switching.py:1019-1060
    def get_step(self):
        return self.getGlobalVariableByName("step")

    def reset(self):
        if self._engine is not None:
            self._engine.reset()

    def getStatistics(self, context=None):
        if self.has_statistics:
            return (self.getGlobalVariableByName("naccept"), self.getGlobalVariableByName("ntrials"))
        return (0, 0)

    def getTotalWork(self, context=None):
        """accumulated total work in units of kT"""
        return self.getGlobalVariableByName("total_work")

    def getShadowWork(self, context=None):
        return self.getGlobalVariableByName("shadow_work")

    def getProtocolWork(self, context=None):
        return self.getGlobalVariableByName("protocol_work")

    def getLogAcceptanceProbability(self, context=None):
        return -1.0 * self.getGlobalVariableByName("total_work")


class NCMCVVAlchemicalIntegrator(NCMCAlchemicalIntegrator):
    """Velocity-Verlet switching (reference blues/switching.py:1083-1241): same signature and defaults."""

    _mode = SWITCH_VV

    def __init__(self, temperature, system, functions, nsteps=0, steps_per_propagation=1, timestep=0.001, direction='insert'):
        super(NCMCVVAlchemicalIntegrator, self).__init__(temperature, system, functions, nsteps, steps_per_propagation, timestep, direction)


class NCMCGHMCAlchemicalIntegrator(NCMCAlchemicalIntegrator):
    """GHMC switching (reference blues/switching.py:1244-1360): same signature and defaults (collision rate 9.1/ps)."""

    _mode = SWITCH_GHMC

    def __init__(self, temperature, system, functions, nsteps=0, steps_per_propagation=1, collision_rate=9.1, timestep=0.001, direction='insert'):
        super(NCMCGHMCAlchemicalIntegrator, self).__init__(temperature, system, functions, nsteps, steps_per_propagation, timestep, direction)
        self._collision_rate = unit.value_in(collision_rate, "1/picosecond")
        self.has_statistics = self.nsteps > 0   # (the reference sets `hasStatistics` in addGHMCStep, switching.py:967)
