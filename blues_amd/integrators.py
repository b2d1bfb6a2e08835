"""Host-side mirror of blues.integrators.AlchemicalExternalLangevinIntegrator
(reference blues/integrators.py:8-249) and of SimulationFactory.generateNCMCIntegrator
(reference blues/simulation.py:650-705).

The object carries the same constructor arguments, defaults and attributes the
reference exposes (`_n_steps_neq`, `_n_lambda_steps`, `_prop_lambda`, `_splitting`,
`_alchemical_functions`, `kT`, `getTemperature`, `getStepSize`, `getGlobalVariableByName`,
`setGlobalVariableByName`, `get_protocol_work`, `getLogAcceptanceProbability`, `reset`,
`step`) but holds no arithmetic: once bound to a context it forwards to the C-ABI engine.
"""
import numpy as np

from . import lepton, unit
from ._abi import IntegratorData

DEFAULT_ALCHEMICAL_FUNCTIONS = {
    # reference blues/simulation.py:654-659
    'lambda_sterics': 'min(1, (1/0.3)*abs(lambda-0.5))',
    'lambda_electrostatics': 'step(0.2-lambda) - 1/0.2*lambda*step(0.2-lambda) + 1/0.2*(lambda-0.8)*step(lambda-0.8)',
}
KB = 0.0083144626  # kJ/mol/K


def get_prop_lambda(prop_lambda):
    """reference blues/integrators.py:147-157"""
    prop_lambda_max = round(prop_lambda + 0.5, 4)
    prop_lambda_min = round(0.5 - prop_lambda, 4)
    if prop_lambda_max - prop_lambda_min <= 0.0:
        prop_lambda_min, prop_lambda_max = 2.0, -1.0
    return prop_lambda_min, prop_lambda_max


class AlchemicalExternalLangevinIntegrator(object):
    """Same signature and defaults as reference blues/integrators.py:98-111."""

    def __init__(self, alchemical_functions, splitting="R V O H O V R", temperature=298.0, collision_rate=1.0,
                 timestep=0.001, constraint_tolerance=1e-8, measure_shadow_work=False, measure_heat=True,
                 nsteps_neq=100, nprop=1, prop_lambda=0.3, seed=0, *args, **kwargs):
        if measure_shadow_work:
            raise NotImplementedError("measure_shadow_work=True is not supported (BLUES never enables it)")
        self._alchemical_functions = dict(alchemical_functions)
        unknown = set(self._alchemical_functions) - {"lambda_sterics", "lambda_electrostatics"}
        if unknown:
            raise ValueError("unsupported alchemical parameters: %s" % sorted(unknown))
        self._splitting = splitting
        tokens = splitting.split()
        bad = [t for t in tokens if t not in ("R", "V", "O", "H")]
        if bad:
            raise ValueError("unsupported splitting tokens %s (R, V, O, H only)" % bad)
        self._temperature = unit.value_in(temperature, "kelvin")
        self._collision_rate = unit.value_in(collision_rate, "1/picosecond")
        self._timestep = unit.value_in(timestep, "picosecond")
        self._constraint_tolerance = float(constraint_tolerance)
        self._measure_heat = measure_heat
        self._n_steps_neq = int(nsteps_neq)
        self._n_lambda_steps = self._n_steps_neq * tokens.count("H")
        self._nprop = int(nprop)
        self._prop_lambda = get_prop_lambda(prop_lambda)
        self._seed = int(seed)
        self._engine = None
        self._pre_globals = {}
        for v in self._alchemical_functions.values():
            lepton.compile_expression(v)

    # ---- what BLUES reads
    @property
    def kT(self):
        return unit.Quantity(KB * self._temperature, "kilojoule/mole")

    def getTemperature(self):
        return unit.Quantity(self._temperature, "kelvin")

    def getStepSize(self):
        return unit.Quantity(self._timestep, "picosecond")

    def setRandomNumberSeed(self, seed):
        self._seed = int(seed)

    def getRandomNumberSeed(self):
        return self._seed

    def to_data(self, replica=0, precision=0):
        n = self._n_lambda_steps
        ls = lepton.tabulate(self._alchemical_functions.get("lambda_sterics", "1"), n)
        le = lepton.tabulate(self._alchemical_functions.get("lambda_electrostatics", "1"), n)
        return IntegratorData(timestep=self._timestep, temperature=self._temperature, nsteps_neq=self._n_steps_neq,
                              lambda_sterics=np.array(ls), lambda_electrostatics=np.array(le), splitting=self._splitting,
                              collision_rate=self._collision_rate, nprop=self._nprop, prop_lambda_min=self._prop_lambda[0],
                              prop_lambda_max=self._prop_lambda[1], constraint_tolerance=self._constraint_tolerance,
                              seed=self._seed, replica=replica, precision=precision)

    # ---- engine-backed calls
    def _bind(self, engine):
        self._engine = engine
        for k, v in self._pre_globals.items():
            engine.set_global(k, v)

    def _need(self):
        if self._engine is None:
            raise RuntimeError("integrator is not bound to a context")
        return self._engine

    def step(self, n):
        self._need().step(int(n))

    def getGlobalVariableByName(self, name):
        if self._engine is None:
            defaults = {"lambda": 0.0, "step": 0.0, "protocol_work": 0.0, "shadow_work": 0.0, "nsteps": self._n_steps_neq,
                        "n_lambda_steps": self._n_lambda_steps, "nprop": self._nprop, "prop": 1.0,
                        "prop_lambda_min": self._prop_lambda[0], "prop_lambda_max": self._prop_lambda[1]}
            defaults.update(self._pre_globals)
            return defaults[name]
        return self._engine.get_global(name)

    def setGlobalVariableByName(self, name, value):
        if self._engine is None:
            self._pre_globals[name] = float(value)
        else:
            self._engine.set_global(name, float(value))

    def get_protocol_work(self, dimensionless=False):
        w = self.getGlobalVariableByName("protocol_work")
        if dimensionless:
            return w / (KB * self._temperature)
        return unit.Quantity(w, "kilojoule/mole")

    def getLogAcceptanceProbability(self, context=None):
        """reference blues/integrators.py:233-238"""
        protocol = self.getGlobalVariableByName("protocol_work")
        shadow = self.getGlobalVariableByName("shadow_work")
        return -1.0 * (protocol + shadow) / (KB * self._temperature)

    def reset(self):
        """reference blues/integrators.py:240-249"""
        if self._engine is not None:
            self._engine.reset()
        self._pre_globals = {}


def generateNCMCIntegrator(nstepsNC=None, alchemical_functions=None, splitting="H V R O R V H", temperature=300.0,
                           dt=0.002, nprop=1, propLambda=0.3, **kwargs):
    """SimulationFactory.generateNCMCIntegrator (reference blues/simulation.py:650-705).
    `friction` in kwargs is swallowed exactly as the reference does: the NCMC collision
    rate is always 1/ps."""
    if alchemical_functions is None:
        alchemical_functions = dict(DEFAULT_ALCHEMICAL_FUNCTIONS)
    return AlchemicalExternalLangevinIntegrator(
        alchemical_functions=alchemical_functions, splitting=splitting, temperature=temperature, nsteps_neq=nstepsNC,
        timestep=dt, nprop=nprop, prop_lambda=propLambda, seed=kwargs.get("seed", 0))


class LangevinIntegrator(object):
    """openmm.LangevinIntegrator(temperature, frictionCoeff, stepSize) stand-in for the MD leg
    (reference blues/simulation.py:629-648).  Splitting token 'L' = OpenMM's Langevin step:
    v' = a v + (1-a)/gamma f/m + sqrt(kT(1-a^2)/m) xi ; x' = x + dt v' ; constrain ; v = (x'-x)/dt."""

    def __init__(self, temperature, frictionCoeff, stepSize, seed=0):
        self._temperature = unit.value_in(temperature, "kelvin")
        self._collision_rate = unit.value_in(frictionCoeff, "1/picosecond")
        self._timestep = unit.value_in(stepSize, "picosecond")
        self._seed = int(seed)
        self._engine = None
        self._constraint_tolerance = 1e-5  # OpenMM's LangevinIntegrator default

    @property
    def kT(self):
        return unit.Quantity(KB * self._temperature, "kilojoule/mole")

    def getTemperature(self): return unit.Quantity(self._temperature, "kelvin")
    def getStepSize(self): return unit.Quantity(self._timestep, "picosecond")
    def getFriction(self): return unit.Quantity(self._collision_rate, "1/picosecond")
    def setRandomNumberSeed(self, seed): self._seed = int(seed)
    def getRandomNumberSeed(self): return self._seed
    def setConstraintTolerance(self, tol): self._constraint_tolerance = float(tol)

    def to_data(self, replica=0, precision=0):
        return IntegratorData(timestep=self._timestep, temperature=self._temperature, nsteps_neq=2 ** 30, lambda_sterics=np.array([1.0]),
                              lambda_electrostatics=np.array([1.0]), splitting="L", collision_rate=self._collision_rate, nprop=1,
                              prop_lambda_min=2.0, prop_lambda_max=-1.0, constraint_tolerance=self._constraint_tolerance,
                              seed=self._seed, replica=replica, precision=precision)

    def _bind(self, engine): self._engine = engine

    def step(self, n):
        if self._engine is None:
            raise RuntimeError("integrator is not bound to a context")
        self._engine.step(int(n))

    def getGlobalVariableByName(self, name):
        return self._engine.get_global(name)
