"""Host-side driver of one BLUES iteration around the native engine: the NCMC leg and the
Metropolis step of blues.simulation.BLUESSimulation (reference blues/simulation.py:812-1257).

Method names, state-table layout, hook order and error policy follow the reference so that
its tests read the same; the difference is below `_ncmc_sim.step`: the reference pays one
Python->OpenMM round trip and ~40 CustomIntegrator sub-steps per NCMC step
(simulation.py:1066-1082), here the steps between two move hooks are one device-resident
`step(n)` call.
"""
import logging
import math
import sys

import numpy as np

from . import unit

logger = logging.getLogger(__name__)
rtol = 6  # reference blues/simulation.py:26-27


class SimulationSet(object):
    """The `simulations` object BLUESSimulation expects: attributes md, alch, ncmc
    (reference blues/simulation.py:768-809).  md / alch may be None: the MD leg is then skipped
    and the alchemical-correction energies (reference simulation.py:1100-1119) are taken from the NCMC engine at lambda = 1.
    In the direct-space model (BLUES_NB_PME_DIRECT) that IS the MD potential.  With reciprocal space (BLUES_NB_PME) it is not: the
    alchemical System keeps the ligand's charges out of the mesh, the self term and the excluded-pair corrections, and its epsilons
    out of the dispersion correction (alchemical_pme_treatment='direct-space', disable_alchemical_dispersion_correction=True), so
    E(lambda = 1) of that System differs from the MD System's energy by the ligand's reciprocal-space and long-range terms and the
    correction comes out near zero instead of the reference's value: hand in `alch` (and `md`) Simulations built from the
    NON-alchemical System, as the reference does, or accept the warning below (throughput runs)."""

    def __init__(self, ncmc, md=None, alch=None):
        self.ncmc, self.md, self.alch = ncmc, md, alch
        system = getattr(ncmc, "system", None)
        if alch is None and getattr(system, "nonbonded_method", None) == 2 and len(getattr(system, "alchemical_atoms", [])):
            logger.warning("SimulationSet without an `alch` Simulation on a PME System: the alchemical correction will use E(lambda=1) of the "
                           "alchemical System, which lacks the ligand's reciprocal-space, self, excluded-pair and dispersion terms")


# ---- the pieces of one BLUES iteration as plain functions over a chain's three Simulations.  BLUESSimulation (one chain) and
# BatchedBLUESSimulation (R chains sharing launches) are both thin drivers over these: the reference's method names are the
# public surface (blues/simulation.py:812-1257 is the contract), the formulation is this repo's.
_STATE_FIELDS = (('positions', 'getPositions', {'asNumpy': True}), ('velocities', 'getVelocities', {'asNumpy': True}),
                 ('potential_energy', 'getPotentialEnergy', {}), ('kinetic_energy', 'getKineticEnergy', {}),
                 ('box_vectors', 'getPeriodicBoxVectors', {}))
_STATE_ORDER = ('box_vectors', 'positions', 'velocities')      # the order the reference writes a State back in (simulation.py:938-963)
_CONTEXT_SETTERS = {'box_vectors': lambda ctx, v: ctx.setPeriodicBoxVectors(*v), 'positions': lambda ctx, v: ctx.setPositions(v),
                    'velocities': lambda ctx, v: ctx.setVelocities(v)}
INTEGRATOR_KEYS = ('lambda', 'shadow_work', 'protocol_work', 'Eold', 'Enew')


def read_state(context, state_keys):
    """A State of `context` as the dict BLUES keeps in its state table (reference blues/simulation.py:883-911)."""
    state = context.getState(**state_keys)
    return {name: getattr(state, getter)(**kw) for name, getter, kw in _STATE_FIELDS}


def write_state(context, table_entry, parts=_STATE_ORDER):
    """The named parts of a state-table entry back into `context` (reference blues/simulation.py:938-963)."""
    for part in _STATE_ORDER:
        if part in parts:
            _CONTEXT_SETTERS[part](context, table_entry[part])
    return context


def metropolis(chain, unmodified_at_x1=None, correction=None):
    """The NCMC acceptance test of one chain (reference blues/simulation.py:1121-1140): log-acceptance of the protocol work plus
    the alchemical correction against the log of a uniform draw.  Returns the decision record; changes nothing.
    unmodified_at_x1: the `alch` context's energy at the switched coordinates where a batch has evaluated it for all chains.
    correction: the correction itself where a batch has formed it as a differential (BatchedBLUESSimulation._decide_batched)."""
    integrator = chain._ncmc_sim.context._integrator
    log_p = integrator.getLogAcceptanceProbability(chain._ncmc_sim.context)
    log_u = math.log(chain._rng.random_sample())
    given, correction = correction, 0.0
    if not np.isnan(log_p):       # a NaN work rejects without asking for energies (the reference's guard)
        correction = chain._computeAlchemicalCorrection(unmodified_at_x1) if given is None else float(given)
        logger.debug('NCMCLogAcceptanceProbability = %.6f + Alchemical Correction = %.6f' % (log_p, correction))
        log_p = log_p + correction
    return {'accept': bool(log_p > log_u), 'log_accept': float(log_p), 'correction': float(correction), 'randnum': log_u,
            'protocol_work': integrator.getGlobalVariableByName('protocol_work')}


def record_decision(chain, decision):
    """Acceptance counters and log line of one decision; returns which state-table entry has to be written where:
    (simulation, entry, parts) or None."""
    chain.last = decision
    verdict = 'ACCEPTED' if decision['accept'] else 'REJECTED'
    logger.info('NCMC MOVE %s: work_ncmc %s %s randnum %s' % (verdict, decision['log_accept'], '>' if decision['accept'] else '<', decision['randnum']))
    if decision['accept']:
        chain.accept += 1
        # the switched configuration becomes the MD state; the MD velocities are redrawn in _resetSimulations anyway
        return (chain._md_sim, chain.stateTable['ncmc']['state1'], ('box_vectors', 'positions')) if chain._md_sim is not None else None
    chain.reject += 1
    if chain._md_sim is not None:
        return None      # the MD context was never touched: nothing to undo (checked in _acceptRejectMove, as the reference does)
    return (chain._ncmc_sim, chain.stateTable['ncmc']['state0'], _STATE_ORDER)   # one context for both legs: undo the switch in place


class BLUESSimulation(object):
    """One BLUES chain: `run` = nIter x (sync MD -> NCMC, NCMC switch with the Move's hooks, Metropolis test, reset, MD leg)."""

    def __init__(self, simulations, config=None, move_engine=None, rng=None, differential_correction=None):
        """differential_correction: None (default) = form the alchemical correction (reference blues/simulation.py:1100-1119) from the
        terms in which the `alch` System and the NCMC System at lambda = 1 differ, evaluated in the NCMC engine, wherever the two
        Systems are recognisably one force field (systems.alchemical_difference_plan) -- SURVEY.md 8f.3; the reference's four total
        energies otherwise.  False: always the four energies (the form the differential is tested against).  True: insist."""
        # the reference draws from numpy's global stream (simulation.py:1133); a private RandomState can be handed in
        # where several chains are driven from worker threads (BatchedBLUESSimulation)
        self._rng = np.random if rng is None else rng
        self._diff_plan = None
        if differential_correction is not False and simulations.alch is not None:
            from . import systems
            a, p = getattr(simulations.ncmc, "system", None), getattr(simulations.alch, "system", None)
            m = getattr(simulations.md, "system", None) if simulations.md is not None else p
            try:
                plan = systems.alchemical_difference_plan(a, p) if hasattr(a, "n_atoms") and hasattr(p, "n_atoms") else None
                if plan is not None and m is not p and systems.alchemical_difference_plan(a, m) is None:
                    plan = None        # (the MD System is another force field than the alch one: U_md(x0) is then not U_alch(x0))
            except Exception:
                plan = None
            eng = getattr(simulations.ncmc.context, "_engine", None)
            if plan is not None and plan["kind"] == "pme" and not hasattr(eng, "mesh_energy"):
                plan = None
            self._diff_plan = plan
            if differential_correction is True and plan is None:
                raise ValueError("differential_correction=True, but the alch / md Systems are not the NCMC System's force field without its alchemical atoms")
        self._dU0 = None
        self._move_engine = move_engine if move_engine is not None else getattr(simulations, "_move_engine", None)
        self._md_sim, self._alch_sim, self._ncmc_sim = simulations.md, simulations.alch, simulations.ncmc
        self._config = config or {}
        self.accept = self.reject = self.acceptRatio = self.currentIter = 0
        self.stateTable = {leg: {'state0': {}, 'state1': {}} for leg in ('md', 'ncmc')}
        self._integrator_keys_ = list(INTEGRATOR_KEYS)
        self._state_keys = dict(getPositions=True, getVelocities=True, getForces=False, getEnergy=True, getParameters=True, enforcePeriodicBox=True)
        self.last = {}

    # ---- the reference's class-level helpers (blues/simulation.py:883-963), kept as names
    @classmethod
    def getStateFromContext(cls, context, state_keys):
        return read_state(context, state_keys)

    @classmethod
    def getIntegratorInfo(cls, ncmc_integrator, integrator_keys=INTEGRATOR_KEYS):
        return {key: ncmc_integrator.getGlobalVariableByName(key) for key in integrator_keys}

    @classmethod
    def setContextFromState(cls, context, state, box=True, positions=True, velocities=True):
        wanted = [p for p, on in zip(_STATE_ORDER, (box, positions, velocities)) if on]
        return write_state(context, state, wanted)

    def _setStateTable(self, simkey, stateidx, stateinfo):
        self.stateTable[simkey][stateidx] = stateinfo

    def _lambda_one_energy(self):
        """U(x; lambda = 1) of the NCMC context as a Quantity: the MD potential where no MD / alchemical context was handed in."""
        return unit.Quantity(self._energy_at_lambda_one(), "kilojoule/mole")

    def _energy_at_lambda_one(self):
        ctx = self._ncmc_sim.context
        eng = getattr(ctx, "_engine", None)
        if eng is not None and hasattr(eng, "potential_energy_at"):
            return eng.potential_energy_at(1.0, 1.0)   # one call; the value normally comes with the previous evaluation
        saved = {p: ctx.getParameter(p) for p in ("lambda_sterics", "lambda_electrostatics")}
        for p in saved:
            ctx.setParameter(p, 1.0)
        e = ctx.getState(getEnergy=True).getPotentialEnergy()._value
        for p, v in saved.items():
            ctx.setParameter(p, v)
        return e

    # ---- D(x) = U_alch(x) - U_ncmc(x; lambda = 1) from the terms in which the two Systems differ
    def _alchemical_difference(self):
        """D at the NCMC context's current coordinates (kJ/mol).  Zero without reciprocal space; with PME the ligand's share of the
        mesh energy (two launches of the NCMC engine's own mesh kernel), the erf corrections of the excluded pairs that hold an
        alchemical atom (host arithmetic on a handful of atoms), and constants of the box."""
        plan = self._diff_plan
        if plan["kind"] == "zero":
            return 0.0
        from . import systems
        ctx = self._ncmc_sim.context
        eng = ctx._engine
        mesh = eng.mesh_energy(True) - eng.mesh_energy(False)
        xyz = ctx.getState(getPositions=True).getPositions(asNumpy=True)[[int(i) for i in plan["atoms"]]]
        xyz = np.asarray(getattr(xyz, "_value", xyz), dtype=np.float64)
        return float(mesh + plan["const"] + systems.excluded_pair_term(plan, xyz))

    # ---- reference blues/simulation.py:1028-1037
    def _syncStatesMDtoNCMC(self):
        if self._md_sim is None:
            # one context serves both legs: its own state is the MD state, the MD potential its energy at lambda = 1
            entry = read_state(self._ncmc_sim.context, self._state_keys)
            entry['potential_energy'] = self._lambda_one_energy()
            self._setStateTable('md', 'state0', entry)
            return
        entry = read_state(self._md_sim.context, self._state_keys)
        self._setStateTable('md', 'state0', entry)
        self._ncmc_sim.context = write_state(self._ncmc_sim.context, entry)
        if self._diff_plan is not None:
            self._dU0 = self._alchemical_difference()      # (at x0, while the NCMC context holds it)

    # ---- reference blues/simulation.py:1039-1098
    def _ncmc_plan(self, nstepsNC, moveStep, move_engine=None):
        """The body of _stepNCMC as a generator: runs the hooks and YIELDS the number of integrator steps to take next;
        whoever drives it advances the NCMC simulation by that many steps (alone: _stepNCMC below; for several replicas
        sharing launches: BatchedBLUESSimulation) and throws a stepping error back in, where the reference's policy applies."""
        logger.info('Advancing %i NCMC switching steps...' % (nstepsNC))
        ncmc_state0 = self.getStateFromContext(self._ncmc_sim.context, self._state_keys)
        self._setStateTable('ncmc', 'state0', ncmc_state0)
        if not move_engine:
            move_engine = self._move_engine
        self._ncmc_sim.currentIter = self.currentIter
        move_engine.selectMove()
        nstepsNC = int(nstepsNC)
        # the reference steps one at a time and tests `step == 0`, `step == moveStep`, `step == lastStep`
        # around each integrator step; the same hook order with the steps in between fused:
        # (the reference's loop never moves and never steps past nstepsNC when moveStep lies outside [0, nstepsNC))
        moveStep = int(moveStep)
        cuts = sorted(set([0, nstepsNC] + ([moveStep] if 0 <= moveStep < nstepsNC else [])))
        try:
            for a, b in zip(cuts[:-1], cuts[1:]):
                if a == 0:
                    self._ncmc_sim.context = move_engine.selected_move.beforeMove(self._ncmc_sim.context)
                if a == moveStep:
                    logger.info('Performing %s...' % move_engine.move_name)
                    self._ncmc_sim.context = move_engine.runEngine(self._ncmc_sim.context)
                if b > a:
                    yield b - a
                if b == nstepsNC:
                    self._ncmc_sim.context = move_engine.selected_move.afterMove(self._ncmc_sim.context)
        except Exception as e:  # reference policy: log, let the move clean up, abandon the switch
            import traceback
            traceback.print_tb(e.__traceback__)
            logger.error(e)
            move_engine.selected_move._error(self._ncmc_sim.context)
        ncmc_state1 = self.getStateFromContext(self._ncmc_sim.context, self._state_keys)
        self._setStateTable('ncmc', 'state1', ncmc_state1)

    def _stepNCMC(self, nstepsNC, moveStep, move_engine=None):
        plan = self._ncmc_plan(nstepsNC, moveStep, move_engine)
        try:
            n = next(plan)
            while True:
                try:
                    self._ncmc_sim.step(n)
                except Exception as e:
                    n = plan.throw(e)
                else:
                    n = plan.send(None)
        except StopIteration:
            pass

    # ---- reference blues/simulation.py:1100-1119: -[ (U_ncmc - U_md)(x0) + (U_alch - U_ncmc)(x1) ] / kT
    def _computeAlchemicalCorrection(self, unmodified_at_x1=None):
        if self._diff_plan is not None and unmodified_at_x1 is None and self._dU0 is not None and self._md_sim is not None:
            # (U_ncmc - U_md)(x0) = -D(x0), (U_alch - U_ncmc)(x1) = +D(x1): -[ -D(x0) + D(x1) ] / kT.  The NCMC context still holds x1.
            return (self._dU0 - self._alchemical_difference()) / self._ncmc_sim.context._integrator.kT._value
        table = self.stateTable
        at_x0 = table['ncmc']['state0']['potential_energy'] - table['md']['state0']['potential_energy']
        end = table['ncmc']['state1']
        if unmodified_at_x1 is not None:
            pass       # (evaluated for the whole batch: BatchedBLUESSimulation._decide_batched)
        elif self._alch_sim is None:
            unmodified_at_x1 = self._lambda_one_energy()
        else:
            self._alch_sim.context = write_state(self._alch_sim.context, end, ('box_vectors', 'positions'))
            unmodified_at_x1 = self._alch_sim.context.getState(getEnergy=True).getPotentialEnergy()
        return (at_x0 + unmodified_at_x1 - end['potential_energy']) * (-1.0 / self._ncmc_sim.context._integrator.kT)

    # ---- reference blues/simulation.py:1121-1166
    def _acceptRejectMove(self, write_move=False):
        todo = record_decision(self, metropolis(self))
        if todo is not None:
            sim, entry, parts = todo
            sim.context = write_state(sim.context, entry, parts)
        elif self._md_sim is not None:
            # rejected with a separate MD context: it must still be where the iteration started (the reference's sanity check)
            before = self.stateTable['md']['state0']['potential_energy']
            now = self._md_sim.context.getState(getEnergy=True).getPotentialEnergy()
            if not math.isclose(before._value, now._value, rel_tol=10.0 ** -rtol):
                logger.error('Last MD potential energy %s != Current MD potential energy %s. Potential energy should match the prior state.' % (before, now))
                sys.exit(1)

    # ---- reference blues/simulation.py:1168-1187 (the `temperature` quirk kept: run() passes its own default of 300)
    def _resetSimulations(self, temperature=None):
        integrator = self._ncmc_sim.context._integrator
        self._ncmc_sim.currentStep = 0
        integrator.reset()
        leg = self._ncmc_sim if self._md_sim is None else self._md_sim
        leg.context.setVelocitiesToTemperature(temperature or integrator.getTemperature(), self._rng.randint(0, 2 ** 31 - 1))

    # ---- reference blues/simulation.py:1189-1213: an MD failure ends the run
    def _stepMD(self, nstepsMD):
        if self._md_sim is None or not nstepsMD:
            return
        logger.info('Advancing %i MD steps...' % (nstepsMD))
        self._md_sim.currentIter = self.currentIter
        try:
            self._md_sim.step(int(nstepsMD))
        except Exception as e:
            logger.error(e, exc_info=True)
            sys.exit(1)

    # ---- reference blues/simulation.py:1215-1257
    def run(self, nIter=0, nstepsNC=0, moveStep=0, nstepsMD=0, temperature=300, write_move=False, **config):
        cfg = self._config
        nIter, nstepsNC, moveStep = int(nIter or cfg['nIter']), nstepsNC or cfg['nstepsNC'], moveStep or cfg['moveStep']
        nstepsMD = nstepsMD or cfg.get('nstepsMD', 0)
        logger.info('Running %i BLUES iterations...' % (nIter))
        for self.currentIter in range(nIter):
            logger.info('BLUES Iteration: %s' % self.currentIter)
            self._syncStatesMDtoNCMC()
            self._stepNCMC(nstepsNC, moveStep)
            self._acceptRejectMove(write_move)
            self._resetSimulations(temperature)
            self._stepMD(nstepsMD)
        self.acceptRatio = self.accept / float(nIter)
        logger.info('Acceptance Ratio: %s' % self.acceptRatio)
        logger.info('nIter: %s ' % nIter)


class BatchedBLUESSimulation(object):
    """R independent BLUES chains on ONE GPU advanced in lock step (DESIGN.md "Replica batches").

    Every chain is an ordinary BLUESSimulation (own SimulationSet, MoveEngine, state table, acceptance counters) --
    exactly what R separate reference processes would hold.  The only thing shared is the integrator stepping: the
    NCMC engines form one native batch (and the MD engines another), so `step(n)` is one kernel launch sequence for all
    chains.  Hooks, state exchange and the Metropolis test run per chain, in chain order."""

    def __init__(self, chains, workers=1, batched_boundary=True, device_turn=None, isolate_failures=False):
        """workers > 1: the per-chain host work (hooks, state exchange through the plugin boundary, Metropolis test) of
        different chains runs on a thread pool -- the C-ABI calls release the GIL and engines are independent objects;
        each chain then draws from its own RandomState (seeded here, in chain order, from numpy's global stream).

        device_turn: a threading.Lock shared by several BatchedBLUESSimulation objects on ONE GPU, each driven from its own
        host thread.  Their stepping calls then take turns on the device (one batch's kernels have the GPU to themselves),
        and the per-chain host work of one batch -- a tenth of an iteration's wall time -- runs while another batch steps.

        isolate_failures: the reference runs one chain per process, so a chain that dies -- an exception from getState after a switch
        that blew up (blues/simulation.py:1096 is outside the try), sys.exit(1) after an MD-leg exception (:1203-1213) -- takes nobody
        with it.  True gives the chains of a batch the same independence on the batched path: the chain is logged and retired (`dead`:
        {chain index: the exception}), sits every later operation out, and the others carry on with the results they would have had
        anyway (chains share launches, never data; every chain that still draws from numpy's global stream is given a private RandomState
        here, seeded from it in chain order, so that a retired chain's missing draws shift nobody's numbers).  Covered: the State reads at
        the head of the iteration and after the switch, the `alch` energy of the correction, the MD energy check after a rejection, the MD
        leg; on the chain-by-chain path (reporters on the NCMC leg, batched_boundary=False) a retired chain is skipped by every per-chain
        operation.  False (default): the first such failure ends the run, as ONE reference process ends."""
        from .engine import NativeBatch
        self.isolate_failures = bool(isolate_failures)
        self.dead = {}
        self.chains = list(chains)
        if not self.chains:
            raise ValueError("no chains")
        self.batched_boundary = bool(batched_boundary)   # False: hooks, State hand-overs and the Metropolis step chain by chain, always
        self._pool = None
        if workers and workers > 1 and len(self.chains) > 1:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=min(int(workers), len(self.chains)))
            for c in self.chains:
                if c._rng is np.random:
                    c._rng = np.random.RandomState(np.random.randint(0, 2 ** 31 - 1))
        if self.isolate_failures:
            # "the others carry on with the results they would have had anyway" needs every chain on a stream of its own: chains that share
            # numpy's global stream (rng=None, the reference's default) would see shifted uniforms after a retired chain's skipped draws
            for c in self.chains:
                if c._rng is np.random:
                    c._rng = np.random.RandomState(np.random.randint(0, 2 ** 31 - 1))
        self._ncmc_batch = NativeBatch([c._ncmc_sim.context._engine for c in self.chains])
        self._ncmc_batch.device_turn = device_turn
        self._md_batch = None
        # (a MonteCarloBarostat on the MD leg -- reference blues/simulation.py:603-626 -- leaves every chain in its own box: the batch's
        # argument records carry the box per member, and the attempts are made member by member where the chunks are planned)
        if all(c._md_sim is not None for c in self.chains):
            self._md_batch = NativeBatch([c._md_sim.context._engine for c in self.chains])
            self._md_batch.device_turn = device_turn
        elif any(c._md_sim is not None for c in self.chains):
            raise ValueError("either every chain has an MD simulation or none has")
        self._alch_batch = None
        self._dU0_all = None
        if all(c._alch_sim is not None for c in self.chains):
            self._alch_batch = NativeBatch([c._alch_sim.context._engine for c in self.chains])
        elif any(c._alch_sim is not None for c in self.chains):
            raise ValueError("either every chain has an `alch` simulation or none has")

    def close(self):
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None
        for b in (self._ncmc_batch, self._md_batch, self._alch_batch):
            if b is not None:
                b.close()

    def for_each_chain(self, fn, indices=None):
        """[fn(r, chain) for the given chains] -- on the pool when there is one."""
        idx = list(range(len(self.chains))) if indices is None else list(indices)
        if self.dead:      # (a retired chain sits every per-chain operation out, on the chain-by-chain path as on the batched one)
            idx = [r for r in idx if r not in self.dead]
        if self._pool is None or len(idx) < 2:
            return [fn(r, self.chains[r]) for r in idx]
        return list(self._pool.map(lambda r: fn(r, self.chains[r]), idx))

    def _alive(self):
        return [r not in self.dead for r in range(len(self.chains))]

    def _retire(self, r, error, where):
        """A chain that cannot go on (isolate_failures): logged with what the reference would have logged, then left out of everything."""
        logger.error("chain %d is retired after a failure in %s: %s" % (r, where, error))
        self.dead[r] = error
        self.chains[r].last = dict(self.chains[r].last or {}, failed=True)

    @staticmethod
    def _advance(batch, sims, wanted):
        """Advance sims[r] by wanted[r] steps (missing key: the member sits this one out).  Returns {r: error}."""
        errors = {}
        left = dict(wanted)
        while left:
            plans = {r: sims[r]._plan_chunk(sims[r].currentStep + left[r]) for r in left}
            n = min(p[0] for p in plans.values())
            errs, _ = batch.step(n, active=[r in left for r in range(len(sims))], raise_errors=False)
            for r in list(left):
                if errs[r] is not None:
                    errors[r] = errs[r]
                    del left[r]
                    continue
                sims[r]._commit_chunk(n, plans[r][1])
                left[r] -= n
                if left[r] <= 0:
                    del left[r]
        return errors

    # ---- the plugin boundary for all chains at once (include/blues_engine.h: blues_batch_snapshot_capture ...).  Chain by chain
    # the hooks, State hand-overs and the Metropolis step are ~60 C-ABI calls and ~25 small launches per chain and iteration;
    # where every chain's Move exposes its geometry as `propose(coordinates of its atoms)` and leaves the other hooks alone, the
    # same sequence of operations is issued once for the whole batch.  Same results, bit for bit, as the chain-by-chain path
    # (tests/test_gpu_batch.py); chains with other Moves take that path.
    def _batchable(self):
        """The State hand-overs, the Metropolis step and the reset of all chains in one call per operation: chains of one shape
        (same System size and box, no barostat, no reporters on the NCMC leg), with or without the md / alch Simulations of the
        reference's triple (blues/simulation.py:768-809).  What the Move's hooks do is a separate question (_move_batchable)."""
        if not self.batched_boundary or not hasattr(self._ncmc_batch, "snapshot_all"):
            return False
        first_sim = None
        for c in self.chains:
            me = c._move_engine
            if me is None or len(getattr(me, "moves", [])) < 1:
                return False
            # one launch per operation is sized and laid out for ONE system: same atoms, same box, no barostat moving it
            if first_sim is None:
                first_sim = c._ncmc_sim
            sysm, sys0 = getattr(c._ncmc_sim, "system", None), getattr(first_sim, "system", None)
            if sysm is not sys0 and (getattr(sysm, "n_atoms", None) != getattr(sys0, "n_atoms", None) or not np.array_equal(getattr(sysm, "box", None), getattr(sys0, "box", None))):
                return False
            if c._ncmc_sim.reporters or getattr(c._ncmc_sim, "barostat", None) is not None:
                return False
            for other in (c._md_sim, c._alch_sim):
                if other is None:
                    continue
                so = getattr(other, "system", None)
                if getattr(other, "barostat", None) is not None or getattr(so, "n_atoms", None) != getattr(sys0, "n_atoms", None) or not np.array_equal(getattr(so, "box", None), getattr(sys0, "box", None)):
                    return False
        return True

    def _move_batchable(self):
        """The Move itself as ONE gather, the chains' propose() calls on the host, ONE scatter: every chain's move is of one class
        that exposes its geometry as `propose(coordinates of its atoms)` on the same atoms and leaves the other hooks alone.  Other
        moves (WaterTranslationMove's beforeMove / afterMove, third-party Move subclasses) run their hooks chain by chain."""
        from . import moves
        first = None
        for c in self.chains:
            me = c._move_engine
            if me is None or len(getattr(me, "moves", [])) != 1:
                return False
            m = me.moves[0]
            if not hasattr(m, "propose") or any(getattr(type(m), hook) is not getattr(moves.Move, hook) for hook in ("beforeMove", "afterMove", "_error")):
                return False
            # the batched path calls propose() in place of move(): only valid where move() is the one written around that propose()
            # (a subclass that overrides move() alone would have its override bypassed)
            owner = lambda name: next(k for k in type(m).__mro__ if name in vars(k))
            if owner("move") is not owner("propose"):
                return False
            if first is None:
                first = m
            elif type(m) is not type(first) or list(m.atom_indices) != list(first.atom_indices):
                return False
        return True

    def _capture_states(self, active=None, leg="ncmc", tolerate=False):
        """getStateFromContext (reference blues/simulation.py:883-911) of every chain's NCMC (or MD) context: one capture for all.
        tolerate: a chain whose context cannot be read (it blew up) yields the exception in place of a State instead of raising."""
        batch = self._ncmc_batch if leg == "ncmc" else self._md_batch
        snaps = batch.snapshot_all(True, True, active=active)
        out = []
        for r, c in enumerate(self.chains):
            if snaps[r] is None:
                out.append(None)
                continue
            ctx = (c._ncmc_sim if leg == "ncmc" else c._md_sim).context
            e = ctx._engine
            try:
                pe, ke = e.energies()
            except Exception as err:
                if not tolerate:
                    raise
                out.append(err)
                continue
            # (enforcePeriodicBox=True of the chain-by-chain path: what reaches the host is wrapped molecule by molecule, context.getState)
            periodic = getattr(ctx._system, "nonbonded_method", 1) != 0 and hasattr(ctx._system, "n_atoms") and hasattr(ctx, "periodic_wrapper")
            out.append({'positions': unit.DeviceQuantity(snaps[r], 1, "nanometer", wrapper=ctx.periodic_wrapper() if periodic else None),
                        'velocities': unit.DeviceQuantity(snaps[r], 2, "nanometer/picosecond"),
                        'potential_energy': unit.Quantity(pe, "kilojoule/mole"), 'kinetic_energy': unit.Quantity(ke, "kilojoule/mole"),
                        'box_vectors': list(ctx.box_vector_quantities()) if hasattr(ctx, "box_vector_quantities") else [unit.Quantity(np.array(row), "nanometer") for row in e.get_box()]})
        return out

    def _restore_states(self, states, velocities=True, leg="ncmc"):
        """setContextFromState of every chain whose entry is not None into its NCMC / MD / alch context (box vectors of an NVT
        chain never change: context.py).  A State captured from one leg's engine goes into another's device to device."""
        batch = {"ncmc": self._ncmc_batch, "md": self._md_batch, "alch": self._alch_batch}[leg]
        sim_of = {"ncmc": (lambda c: c._ncmc_sim), "md": (lambda c: c._md_sim), "alch": (lambda c: c._alch_sim)}[leg]
        snaps = [None if st is None else (st['positions'].on_device() if hasattr(st['positions'], "on_device") else None) for st in states]
        if any(st is not None and sn is None for st, sn in zip(states, snaps)) or not batch.restore_all(snaps, True, velocities):
            for c, st in zip(self.chains, states):   # somebody read or edited the arrays on the host: the member calls
                if st is not None:
                    sim = sim_of(c)
                    sim.context = c.setContextFromState(sim.context, st, velocities=velocities)

    def _stepNCMC_batched(self, nstepsNC, moveStep):
        chains, batch = self.chains, self._ncmc_batch
        R = len(chains)
        sims = [c._ncmc_sim for c in chains]
        alive = self._alive() if self.dead else None
        batch.prefetch_energies(active=alive)
        for c, st in zip(chains, self._capture_states(active=alive)):
            if st is None:
                continue
            c._setStateTable('ncmc', 'state0', st)
            c._ncmc_sim.currentIter = c.currentIter
            c._move_engine.selectMove()
        nstepsNC, moveStep = int(nstepsNC), int(moveStep)
        cuts = sorted(set([0, nstepsNC] + ([moveStep] if 0 <= moveStep < nstepsNC else [])))
        one_scatter = self._move_batchable()
        failed = dict(self.dead)      # (retired chains sit the switch out like chains whose switch has been abandoned)

        def hook(r, fn):
            """A Move hook of chain r under the reference's policy (blues/simulation.py:1088-1094): an exception is logged, the
            move cleans up, the chain's switch is abandoned."""
            c = chains[r]
            try:
                c._ncmc_sim.context = fn(c, c._ncmc_sim.context)
            except Exception as e:
                import traceback
                traceback.print_tb(e.__traceback__)
                logger.error(e)
                c._move_engine.selected_move._error(c._ncmc_sim.context)
                failed[r] = e

        for a, b in zip(cuts[:-1], cuts[1:]):
            if a == 0 and not one_scatter:
                for r in range(R):
                    if r not in failed:
                        hook(r, lambda c, ctx: c._move_engine.selected_move.beforeMove(ctx))
            live = [r not in failed for r in range(R)]
            if a == moveStep and any(live):
                if a > 0:   # the work of the instantaneous move needs U(x) before the edit (integrators.py:184-205): for all chains at once
                    batch.prefetch_energies(active=live, kinetic=False)
                if one_scatter:
                    idx = list(chains[0]._move_engine.selected_move.atom_indices)
                    snaps = batch.snapshot_all(True, False, active=live)                 # context.getState(getPositions=True)
                    xyz = batch.read_atoms_all(idx, snaps=snaps)                         # positions[atom_indices]
                    new = np.array([chains[r]._move_engine.selected_move.propose(xyz[r]) if live[r] else xyz[r] for r in range(R)])
                    if not batch.restore_edited_all(snaps, idx, new):                    # positions[i] = ...; context.setPositions(positions)
                        for r in range(R):
                            if live[r]:
                                q = unit.DeviceQuantity(snaps[r], 1, "nanometer")
                                for k, i in enumerate(idx):
                                    q[i] = new[r][k]
                                chains[r]._ncmc_sim.context.setPositions(q)
                    for r in range(R):
                        if live[r]:
                            m = chains[r]._move_engine.selected_move
                            m.positions = new[r]
                            chains[r]._move_engine.move_name = getattr(chains[r]._move_engine, "move_name", type(m).__name__)
                else:
                    for r in range(R):
                        if live[r]:
                            logger.info('Performing %s...' % getattr(chains[r]._move_engine, "move_name", "move"))
                            hook(r, lambda c, ctx: c._move_engine.runEngine(ctx))
            if b > a:
                for r, e in self._advance(batch, sims, {r: b - a for r in range(R) if r not in failed}).items():
                    logger.error(e)       # reference policy (simulation.py:1088-1094): log, let the move clean up, abandon that chain's switch
                    chains[r]._move_engine.selected_move._error(chains[r]._ncmc_sim.context)
                    failed[r] = e
            if b == nstepsNC and not one_scatter:
                for r in range(R):
                    if r not in failed:
                        hook(r, lambda c, ctx: c._move_engine.selected_move.afterMove(ctx))
        batch.prefetch_energies(active=[r not in failed for r in range(R)], at_lambda_one=True)
        # the reference reads the State of a chain whose switch was abandoned all the same (simulation.py:1096: outside the try); a
        # context that blew up raises there and that chain's process ends.  isolate_failures: that chain is retired, the others go on.
        states = self._capture_states(active=self._alive() if self.dead else None, tolerate=self.isolate_failures)
        for r, (c, st) in enumerate(zip(chains, states)):
            if isinstance(st, Exception):
                self._retire(r, st, "reading the State after the NCMC switch")
            elif st is not None:
                c._setStateTable('ncmc', 'state1', st)

    def _alchemical_difference_all(self, alive):
        """D(x) of every live chain at the coordinates its NCMC context holds now (BLUESSimulation._alchemical_difference for the
        whole batch): two batched launches of the mesh kernel, one gather of the handful of atoms the excluded-pair term needs."""
        plan = self.chains[0]._diff_plan
        R = len(self.chains)
        if plan["kind"] == "zero":
            return np.zeros(R)
        from . import systems
        mask = alive if self.dead else None
        mesh = self._ncmc_batch.mesh_energy_all(True, active=mask) - self._ncmc_batch.mesh_energy_all(False, active=mask)
        xyz = self._ncmc_batch.read_atoms_all([int(i) for i in plan["atoms"]])
        return mesh + plan["const"] + systems.excluded_pair_term(plan, xyz)

    def _differential(self):
        """The alchemical correction of every chain as a differential: every chain has a plan, and it is the same one."""
        plans = [c._diff_plan for c in self.chains]
        if any(p is None for p in plans) or self._md_batch is None or not hasattr(self._ncmc_batch, "mesh_energy_all"):
            return False
        first = plans[0]
        return all(p is first or (p["kind"] == first["kind"] and p.get("const") == first.get("const") and np.array_equal(p.get("atoms"), first.get("atoms"))) for p in plans)

    def _decide_batched(self, temperature):
        """_acceptRejectMove of every chain (reference blues/simulation.py:1121-1166): the `alch` energies of the correction for
        all chains at once, the tests chain by chain on numbers already on the host, then the State write-backs in one call --
        accepted chains into their MD contexts where there are such, rejected ones back in place where there are none."""
        chains = self.chains
        R = len(chains)
        unmodified = [None] * R
        alive = self._alive()
        corrections = [None] * R
        if self._differential() and self._dU0_all is not None:
            # -[ (U_ncmc - U_md)(x0) + (U_alch - U_ncmc)(x1) ] / kT = (D(x0) - D(x1)) / kT with D = U_alch - U_ncmc(lambda = 1): formed in the
            # NCMC engine from the terms in which the two Systems differ -- no copy of the switched coordinates into the `alch` batch,
            # no evaluation of an all-mobile System (a full list rebuild for ONE energy), no frozen-frozen constant
            dU1 = self._alchemical_difference_all(alive)
            for r, c in enumerate(chains):
                if alive[r]:
                    corrections[r] = (self._dU0_all[r] - dU1[r]) / c._ncmc_sim.context._integrator.kT._value
        elif self._alch_batch is not None:
            # reference simulation.py:1107-1110: the switched coordinates into the alch context, its potential energy
            ends = [c.stateTable['ncmc']['state1'] if alive[r] else None for r, c in enumerate(chains)]
            self._restore_states(ends, velocities=False, leg="alch")
            self._alch_batch.prefetch_energies(kinetic=False, active=alive if self.dead else None)
            for r, c in enumerate(chains):
                if not alive[r]:
                    continue
                try:
                    unmodified[r] = c._alch_sim.context.getState(getEnergy=True).getPotentialEnergy()
                except Exception as e:       # (the switched coordinates cannot be evaluated: that chain's process would end here)
                    if not self.isolate_failures:
                        raise
                    self._retire(r, e, "the alch energy of the correction")
                    alive[r] = False
        restore = []
        for r, c in enumerate(chains):
            if not alive[r]:
                c._rng.random_sample()      # (its own stream under isolate_failures: drawn for symmetry with the seed of _reset_batched)
                restore.append(None)
                continue
            todo = record_decision(c, metropolis(c, unmodified[r], correction=corrections[r]))
            restore.append(None if todo is None else todo[1])
        if self._md_batch is None:
            self._restore_states(restore)            # a rejection restores the pre-switch state in place
            return
        self._restore_states(restore, velocities=False, leg="md")     # accepted: the switched configuration becomes the MD state
        # rejected: the MD context must still be where the iteration started (the reference's sanity check, simulation.py:1156-1166):
        # the energies of all such chains in one evaluation, the comparisons on host numbers
        rejected = [st is None and alive[r] for r, st in enumerate(restore)]
        if any(rejected):
            self._md_batch.prefetch_energies(kinetic=False, active=rejected)
        for r, c in enumerate(chains):
            if not rejected[r]:
                continue
            before = c.stateTable['md']['state0']['potential_energy']
            try:
                now = c._md_sim.context.getState(getEnergy=True).getPotentialEnergy()
            except Exception as e:
                if not self.isolate_failures:
                    raise
                self._retire(r, e, "the MD energy check after a rejection")
                continue
            if not math.isclose(before._value, now._value, rel_tol=10.0 ** -rtol):
                logger.error('Last MD potential energy %s != Current MD potential energy %s. Potential energy should match the prior state.' % (before, now))
                sys.exit(1)

    def _reset_batched(self, temperature):
        """_resetSimulations of every chain (reference blues/simulation.py:1168-1187): one reset, one velocity redraw (of the MD
        contexts where the chains have them)."""
        chains = self.chains
        seeds, temps = [], []
        for c in chains:
            c._ncmc_sim.currentStep = 0
            c._ncmc_sim.context._integrator._pre_globals = {}       # integrator.reset(): the engine part follows for all chains at once
            temps.append(unit.value_in(temperature if temperature else c._ncmc_sim.context._integrator.getTemperature(), "kelvin"))
            seeds.append(c._rng.randint(0, 2 ** 31 - 1))
        alive = self._alive()       # (seeds are drawn for every chain, retired or not: the others' streams do not move)
        self._ncmc_batch.reset_all(active=alive if self.dead else None)
        leg = self._ncmc_batch if self._md_batch is None else self._md_batch
        for T in sorted(set(temps)):   # (one launch per distinct temperature: one, in practice)
            leg.set_velocities_to_temperature_all(T, seeds, active=[t == T and ok for t, ok in zip(temps, alive)])

    def _sync_batched(self):
        """_syncStatesMDtoNCMC of every chain (reference blues/simulation.py:1028-1037).  With MD contexts: their States in one
        capture, into the NCMC contexts in one restore (device to device).  Without: the MD state is the NCMC context's own, its
        potential the one at lambda = 1."""
        alive = self._alive() if self.dead else None
        if self._md_batch is not None:
            self._md_batch.prefetch_energies(active=alive)
            # (an MD context that blew up without raising in its leg raises HERE, when its State is read: under isolate_failures that
            # chain is retired like one whose leg raised, reference blues/simulation.py:1203-1213 -- and does not end the batch)
            states = self._capture_states(active=alive, leg="md", tolerate=self.isolate_failures)
            for r, (c, st) in enumerate(zip(self.chains, states)):
                if isinstance(st, Exception):
                    self._retire(r, st, "reading the MD State at the head of the iteration")
                    states[r] = None
                elif st is not None:
                    c._setStateTable('md', 'state0', st)
            self._restore_states(states, velocities=True, leg="ncmc")
            self._dU0_all = self._alchemical_difference_all(self._alive()) if self._differential() else None      # (at x0, while the NCMC contexts hold it)
            return
        self._ncmc_batch.prefetch_energies(at_lambda_one=True, active=alive)
        for r, (c, st) in enumerate(zip(self.chains, self._capture_states(active=alive, tolerate=self.isolate_failures))):
            if isinstance(st, Exception):
                self._retire(r, st, "reading the State at the head of the iteration")
                continue
            if st is None:
                continue
            st['potential_energy'] = c._lambda_one_energy()
            c._setStateTable('md', 'state0', st)

    def _stepNCMC(self, nstepsNC, moveStep, batchable=None):
        """batchable: what _batchable() returned a moment ago, where the caller has just asked (it walks every chain)."""
        if self._batchable() if batchable is None else batchable:
            return self._stepNCMC_batched(nstepsNC, moveStep)
        sims = [c._ncmc_sim for c in self.chains]
        plans = {r: c._ncmc_plan(nstepsNC, moveStep) for r, c in enumerate(self.chains)}
        DONE = object()

        def resume(r, err=None, first=False):
            try:
                if first:
                    return next(plans[r])
                return plans[r].throw(err) if err is not None else plans[r].send(None)
            except StopIteration:
                return DONE

        # every chain is about to ask its context for energies (state0 here; the unperturbed energy of the move and state1
        # after each advance): evaluate them for the whole batch at once, the per-chain calls then find them cached
        live = [r for r in range(len(sims)) if r not in self.dead]
        self._ncmc_batch.prefetch_energies(active=self._alive() if self.dead else None)
        got = self.for_each_chain(lambda r, c: resume(r, first=True), live)
        wanted = {r: n for r, n in zip(live, got) if n is not DONE}
        while wanted:
            errors = self._advance(self._ncmc_batch, sims, wanted)
            idx = sorted(wanted)
            # (at_lambda_one: the energies _computeAlchemicalCorrection and _syncStatesMDtoNCMC take at lambda = 1)
            self._ncmc_batch.prefetch_energies(active=[r in wanted and r not in errors for r in range(len(sims))], at_lambda_one=True)
            got = self.for_each_chain(lambda r, c: resume(r, errors.get(r)), idx)
            wanted = {r: n for r, n in zip(idx, got) if n is not DONE}

    def _stepMD(self, nstepsMD):
        if self._md_batch is None or not nstepsMD:
            return
        sims = [c._md_sim for c in self.chains]
        for r, c in enumerate(self.chains):
            sims[r].currentIter = c.currentIter
        errors = self._advance(self._md_batch, sims, {r: int(nstepsMD) for r in range(len(sims)) if r not in self.dead})
        for r, e in errors.items():  # reference blues/simulation.py:1207-1213: an MD failure is fatal -- to that chain's process
            if self.isolate_failures:
                self._retire(r, e, "the MD leg")
                continue
            logger.error(e, exc_info=True)
            sys.exit(1)

    def run(self, nIter=0, nstepsNC=0, moveStep=0, nstepsMD=0, temperature=300, write_move=False, on_iteration=None, **config):
        """`write_move` is accepted for signature parity and unused, as in the reference (blues/simulation.py:1121: the argument of
        _acceptRejectMove is never read there either)."""
        cfg = self.chains[0]._config
        if not nIter: nIter = cfg['nIter']
        if not nstepsNC: nstepsNC = cfg['nstepsNC']
        if not nstepsMD: nstepsMD = cfg.get('nstepsMD', 0)
        if not moveStep: moveStep = cfg['moveStep']
        for N in range(int(nIter)):
            def sync(r, c):
                c.currentIter = N
                c._syncStatesMDtoNCMC()
            fast = self._batchable()      # (once per iteration: the hooks of an iteration do not change what the chains are)
            if fast:
                for c in self.chains:
                    c.currentIter = N
                self._sync_batched()
            else:
                (self._md_batch or self._ncmc_batch).prefetch_energies(at_lambda_one=self._md_batch is None, active=self._alive() if self.dead else None)
                self.for_each_chain(sync)
            self._stepNCMC(nstepsNC, moveStep, batchable=fast)
            if fast:
                self._decide_batched(temperature)
                if on_iteration is not None:     # (between the decision and the reset, as on the chain-by-chain path below)
                    on_iteration(N, [c.last for c in self.chains])
                self._reset_batched(temperature)
            else:
                self.for_each_chain(lambda r, c: c._acceptRejectMove(write_move))
                if on_iteration is not None:
                    on_iteration(N, [c.last for c in self.chains])
                self.for_each_chain(lambda r, c: c._resetSimulations(temperature))
            self._stepMD(nstepsMD)
        for c in self.chains:
            c.acceptRatio = c.accept / float(nIter)
