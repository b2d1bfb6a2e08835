#!/usr/bin/env python3
"""bench.py -- NCMC switching-leg throughput of the MI355X-native engine.

A "step" is one pass of the hot path over one batch of synthetic input: one complete 1000-step NCMC switch
(BASELINE.json configs[1]) of the S23k box (23,400 atoms, 15-atom alchemical toluene, ~276 mobile atoms emulating
freeze_radius 5 A) for EVERY chain of the batch, driven through the drop-in boundary exactly as BLUESSimulation drives
it: state sync, _stepNCMC with the RandomLigandRotationMove at lambda = 0.5, accept/reject, reset.

metric = ns/day of switching trajectory, whole job: the sum over all independent chains on all ranks (SURVEY.md 8d:
"aggregate over independent replicas").  One chain keeps a few percent of an MI355X busy, so each rank runs
--replicas R chains (default 2048) as --groups G replica batches (default 2: batches of 1024 whose stepping calls take
turns on the device; every kernel launch of a batch covers all its chains, DESIGN.md sections 4d and 5); the single-chain
rate of the same switch (configs[1] to the letter: one replica on one GPU) is measured too and reported beside it as
"single_replica".  Ranks are independent (weak scaling, no data-path collective; one all-gather of the accept records
per switch).  Every chain starts from its own state (--decorrelate steps at set-up; --same-start for the common start of
rounds 1-4).  --md-steps N times the reference's FULL iteration instead (NCMC switch + MD leg on the unfrozen System).

    python bench.py --gpus N --steps K --warmup W [--replicas R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NSTEPS_NC = 1000
DT_PS = 0.004
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md
ALGO_BYTES_PER_ATOM = 36.0     # x,y,z + q,sigma,eps read, fx,fy,fz written, per force evaluation
DECORRELATE_LEG = 25           # steps per leg of the set-up decorrelation (lambda <= DECORRELATE_LEG / nstepsNC throughout)


def build_chains(rank, local_rank, nsteps, workload, R, reciprocal=False, md_steps=0, with_alch=True, setup_threads=None):
    """R independent BLUES chains on this rank's GPU: own integrator (Philox key), context, move engine, state table.
    md_steps > 0: every chain gets the reference's full triple (blues/simulation.py:768-809): the NCMC Simulation on the alchemical
    (frozen) System, an MD Simulation (openmm.LangevinIntegrator, reference simulation.py:647) and an `alch` Simulation, both on
    the UNFROZEN, non-alchemical System."""
    from blues_amd import integrators, moves, simulation, systems
    from blues_amd.context import Simulation
    from blues_amd.replicas import replica_seed, build_in_parallel
    import copy
    if workload == "water":
        # configs[3]: nothing frozen, "backbone" restraints, the alchemical species is one 3-atom water (the first one,
        # reference blues/moves.py:889) and the move is WaterTranslationMove within 2.0 nm of a reference group (here the
        # first toluene stands in for the protein selection)
        base, vel = systems.s23k(frozen=False, restrained=40)
        system = copy.copy(base)
        system.alchemical_atoms = np.array([15, 16, 17], np.int32)
        res = np.asarray(system.residue_of_atom)
        o_idx = [i for i in range(15, system.n_atoms - 2) if res[i] == res[i + 2] and (i == 0 or res[i - 1] != res[i]) and system.mass[i] > 10.0]
        waters = [[i, i + 1, i + 2] for i in o_idx]
        make_move = lambda gid: moves.WaterTranslationMove(waters, np.arange(15), system.mass[:15], radius=2.0)
    elif workload == "rotmove-solute":
        # the mobile region of the reference's freeze_radius (blues/simulation.py:394-480: `(<center> <: d) & !(<solvent>)`): the ligand and the
        # 18 toluenes packed around it -- 285 SOLUTE atoms with bonds, angles, torsions, 1-4 exceptions and C-H constraint clusters -- all water
        # frozen (systems.s23k_solute; the headline's mobile set is the ligand + 87 rigid waters: no bonded term, water triangles only)
        system, vel = systems.s23k_solute(frozen=True)
        lig = np.asarray(system.alchemical_atoms)
        make_move = lambda gid: moves.RandomLigandRotationMove(lig, system.mass[lig], random_state=1000 + gid)
    elif workload == "sidechain":
        # configs[4]: a torsion move on a partially alchemical solute (the methyl group of the first toluene: alchemical-
        # environment exclusions and 1-4 exceptions active), long protocol
        system, vel = systems.s23k(mobile_atoms=275, frozen=True)
        system = copy.copy(system)
        system.alchemical_atoms = np.array([0, 7, 8, 9], np.int32)
        make_move = lambda gid: moves.TorsionRotationMove((1, 0), [7, 8, 9], random_state=1000 + gid)
    else:
        system, vel = systems.s23k(mobile_atoms=275, frozen=True)
        lig = np.asarray(system.alchemical_atoms)
        make_move = lambda gid: moves.RandomLigandRotationMove(lig, system.mass[lig], random_state=1000 + gid)
    if reciprocal:   # nonbondedMethod=PME in full: mesh + self + excluded-pair + dispersion terms
        system = systems.with_reciprocal_space(system)
    md_system = None
    if md_steps > 0:
        md_system = (systems.s23k_solute(frozen=False) if workload == "rotmove-solute" else systems.s23k(frozen=False, restrained=40 if workload == "water" else 0))[0]
        md_system = copy.copy(md_system)
        md_system.alchemical_atoms = np.zeros(0, np.int32)
        if reciprocal:
            md_system = systems.with_reciprocal_space(md_system)
    def make_chain(c):
        gid = rank * R + c   # global chain index
        integ = integrators.generateNCMCIntegrator(nstepsNC=nsteps, dt=DT_PS, temperature=300.0, seed=replica_seed(1234, gid))
        sim = Simulation(None, system, integ, device=local_rank, precision="mixed", replica=gid)
        mover = moves.MoveEngine(make_move(gid))
        md = alch = None
        if md_system is not None:
            md = Simulation(None, md_system, integrators.LangevinIntegrator(300.0, 1.0, DT_PS, seed=replica_seed(4321, gid)), device=local_rank, precision="mixed", replica=gid)
            if with_alch:
                alch = Simulation(None, md_system, integrators.LangevinIntegrator(300.0, 1.0, DT_PS, seed=replica_seed(8765, gid)), device=local_rank, precision="mixed", replica=gid)
        return simulation.BLUESSimulation(simulation.SimulationSet(sim, md=md, alch=alch), {"nstepsNC": nsteps, "moveStep": nsteps // 2, "nIter": 1, "nstepsMD": md_steps}, mover)
    chains = build_in_parallel(make_chain, R, workers=setup_threads)   # (host threads: a chain's set-up is native host work)
    return system, vel, chains


def md_states(chains, x0, v0, batch=None, driver=None, decorrelate=0):
    """What the MD leg hands over at the start of every iteration (reference simulation.py:1028-1037: getStateFromContext
    on the MD context, setContextFromState on the NCMC one).  There is no MD leg in this benchmark, so the hand-over State
    is taken once per chain; like any State of this engine it lives in HBM, i.e. the inputs of the timed region are resident
    on the device.  decorrelate > 0 (set-up time): every chain gets its OWN state first -- velocities from its own seed, that
    many steps of its own switch (lambda <= decorrelate / nstepsNC), integrator.reset() -- instead of the common start."""
    from blues_amd import unit
    own_start = decorrelate > 0 and driver is not None
    if not own_start:     # (with own start states the coordinates are the System's own, set when the engine was made, and the velocities are drawn below: 2 x 2048 uploads of 560 KB saved at set-up)
        for c in chains:
            ctx = c._ncmc_sim.context
            ctx.setPositions(unit.Quantity(x0, "nanometer")); ctx.setVelocities(unit.Quantity(v0, "nanometer/picosecond"))
    if own_start:
        driver._reset_batched(300.0)      # (velocities: each chain draws its own seed from its own stream)
        # legs of DECORRELATE_LEG steps with the integrator reset in between: lambda never passes DECORRELATE_LEG / nstepsNC (0.025: the
        # ligand keeps >= 87 % of its charges, sterics untouched), so the hand-over State is an equilibrium state of the lambda = 0
        # System to that approximation -- not a configuration relaxed around a half-decoupled ligand (one 250-step leg reached 0.25)
        sims = [c._ncmc_sim for c in chains]
        left = int(decorrelate)
        while left > 0:
            n = min(DECORRELATE_LEG, left); left -= n
            errors = driver._advance(driver._ncmc_batch, sims, {r: n for r in range(len(chains))})
            if errors:
                raise RuntimeError("decorrelation leg failed: %s" % list(errors.values())[0])
            for c in chains:
                c._ncmc_sim.currentStep = 0
                c._ncmc_sim.context._integrator._pre_globals = {}
            driver._ncmc_batch.reset_all()
        driver._reset_batched(300.0)
    if batch is not None:      # the energies every chain's State is about to ask for: one evaluation for the whole batch (set-up time only)
        batch.prefetch_energies()
    return [c.getStateFromContext(c._ncmc_sim.context, c._state_keys) for c in chains]


def _record(c, it):
    """[accept, iteration, log_accept, protocol_work, correction] of a chain's last decision (a retired chain: not accepted, NaN)"""
    l = c.last or {}
    if l.get("failed"):
        return [0.0, it, float("nan"), float("nan"), float("nan")]
    return [l["accept"], it, l["log_accept"], l["protocol_work"], l["correction"]]


def one_switch(driver, chains, states, nsteps, it, clock, gather=True):
    """One BLUES iteration's NCMC leg for every chain: MD->NCMC hand-over -> switch -> Metropolis -> gather -> reset."""
    from blues_amd.replicas import gather_decision_block
    each = driver.for_each_chain if driver is not None else (lambda fn: [fn(r, c) for r, c in enumerate(chains)])

    def hand_over(r, c):
        c._ncmc_sim.context = c.setContextFromState(c._ncmc_sim.context, states[r])

    def sync(r, c):
        c.currentIter = it
        c._syncStatesMDtoNCMC()
    fast = driver is not None and driver._batchable()   # the plugin boundary for all chains at once (simulation.py)
    t0 = time.perf_counter()
    if fast:
        driver._restore_states(states)
        for c in chains:
            c.currentIter = it
        driver._sync_batched()
    else:
        each(hand_over)
        if driver is not None:
            driver._ncmc_batch.prefetch_energies(at_lambda_one=True)
        each(sync)
    t1 = time.perf_counter()
    if driver is None:
        chains[0]._stepNCMC(nsteps, nsteps // 2)
    else:
        driver._stepNCMC(nsteps, nsteps // 2, batchable=fast)
    t2 = time.perf_counter()
    if fast:
        driver._decide_batched(300.0)
    else:
        each(lambda r, c: c._acceptRejectMove())
    recs = np.array([_record(c, it) for c in chains], dtype=np.float64)
    if gather:
        recs = gather_decision_block(recs)
    if fast:
        driver._reset_batched(300.0)
    else:
        each(lambda r, c: c._resetSimulations(300.0))
    t3 = time.perf_counter()
    clock["sync"] += t1 - t0; clock["switch"] += t2 - t1; clock["decide"] += t3 - t2
    clock.setdefault("iterations", []).append(t3 - t0)
    _note_layout_events(driver, clock)
    return recs


def _stage(name):
    """One line per stage on stderr (never stdout: the JSON line is alone there): if the process dies -- a GPU memory fault ends it without a
    Python traceback -- the last marker says where."""
    if os.environ.get("RANK", "0") == "0":
        sys.stderr.write("[bench] %s\n" % name); sys.stderr.flush()


def _note_layout_events(driver, clock):
    """What the batch's layout cost in the iteration that just ended (NativeBatch.counters: re-plans of the layout shape, members
    re-sorted at the 64-step polls, members laid out again, the seconds those took): a slow iteration names its cause."""
    if driver is None or not hasattr(driver._ncmc_batch, "counters"):
        return
    now = driver._ncmc_batch.counters()
    last = clock.get("_counters")
    clock["_counters"] = now
    if last is not None:
        clock.setdefault("layout_events", []).append({k: (round(now[k] - last[k], 4) if k.endswith("seconds") else now[k] - last[k])
                                                       for k in ("replans", "replan_seconds", "relayouts", "poll_resorts", "resort_seconds", "reshapes", "straggled", "rejoined", "straggle_seconds", "partial_steps") if k in now})


def one_iteration(driver, chains, nsteps, md_steps, it, clock):
    """One FULL BLUES iteration of every chain, as BLUESSimulation.run does it (reference blues/simulation.py:1215-1257):
    _syncStatesMDtoNCMC -> _stepNCMC -> _acceptRejectMove -> (gather) -> _resetSimulations -> _stepMD on the unfrozen MD System."""
    from blues_amd.replicas import gather_decision_block
    fast = driver._batchable()
    t0 = time.perf_counter()
    for c in chains:
        c.currentIter = it
    if fast:
        driver._sync_batched()
    else:
        (driver._md_batch or driver._ncmc_batch).prefetch_energies(at_lambda_one=driver._md_batch is None)
        driver.for_each_chain(lambda r, c: c._syncStatesMDtoNCMC())
    t1 = time.perf_counter()
    driver._stepNCMC(nsteps, nsteps // 2, batchable=fast)
    t2 = time.perf_counter()
    if fast:
        driver._decide_batched(300.0)
    else:
        driver.for_each_chain(lambda r, c: c._acceptRejectMove())
    recs = gather_decision_block(np.array([_record(c, it) for c in chains], dtype=np.float64))
    if fast:
        driver._reset_batched(300.0)
    else:
        driver.for_each_chain(lambda r, c: c._resetSimulations(300.0))
    t3 = time.perf_counter()
    driver._stepMD(md_steps)
    t4 = time.perf_counter()
    clock["sync"] += t1 - t0; clock["switch"] += t2 - t1; clock["decide"] += t3 - t2; clock["md"] = clock.get("md", 0.0) + t4 - t3
    clock.setdefault("iterations", []).append(t4 - t0)
    _note_layout_events(driver, clock)
    return recs


def cpu_baseline(system, vel, nsteps_sample):
    """The CPU oracle (fp64 restatement of the same step program, 3 full energy/force evaluations per step) on a bounded sample of the
    same workload: (i) one thread, the analogue of OpenMM's single-threaded Reference platform -- the headline baseline; (ii) its pair
    loop on all cores of this host (OpenMP build), core count stated."""
    from blues_amd import integrators
    from oracle import oracle
    integ = integrators.generateNCMCIntegrator(nstepsNC=NSTEPS_NC, dt=DT_PS, temperature=300.0, seed=1234)

    def timed(openmp, nsteps):
        o = oracle.Oracle(system, integ.to_data(), openmp=openmp)
        o.set_velocities(vel)
        o.step(1)  # first-step block + warm caches
        t0 = time.perf_counter()
        o.step(nsteps)
        dt = time.perf_counter() - t0
        return nsteps * DT_PS * 1e-3 / (dt / 86400.0), dt
    v1, dt1 = timed(False, nsteps_sample)
    out = {"value": v1, "unit": "ns/day", "cores": 1, "kind": "port",
           "sample": "%d NCMC steps of the same S23k switch (3 full fp64 energy/force evaluations per step), %.1f s" % (nsteps_sample, dt1)}
    try:
        ncores = int(os.environ.get("OMP_NUM_THREADS", 0)) or oracle.usable_cores()   # (affinity mask capped by the cgroup CPU quota)
        n_omp = max(150, 12 * nsteps_sample)   # about 10-15 s on 16 cores
        vo, dto = timed(True, n_omp)
        out["all_cores"] = {"value": vo, "unit": "ns/day", "cores": ncores, "kind": "port", "sample": "%d steps, %.1f s, OpenMP over the pair loop's cells" % (n_omp, dto)}
    except Exception as e:   # (no OpenMP runtime on the host: the single-thread figure stands alone)
        out["all_cores"] = {"error": str(e)}
    return out


PMC_FILE = "profiles/r06_pmc_nonbonded.json"


def kernel_source_sha():
    """Identifies the build the PMC evidence was taken on (profiles/*pmc*.json carry the same hash): every source of the library AND
    the compiler flags (blues_amd/build.py)."""
    from blues_amd import build
    return build.source_sha()


def pmc_evidence(workload, R):
    """Counters of the batched nonbonded kernel from separate rocprofv3 --pmc passes (scripts/pmc_nb.sh -> profiles/).  They
    cannot be collected inside this run; they are only reported when they were taken on THIS build of the kernel (source
    hash) and this workload / batch size -- otherwise null, never a stale number."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as fh:
            table = json.load(fh)
        e = table.get("%s_R%d" % (workload, R))
        if e and e.get("source_sha") == kernel_source_sha():
            return e
    except Exception:
        pass
    return None


def in_range_counts(system):
    """Host-side census for the byte counts SURVEY.md 8(d) asks for beside the nominal one: environment atoms within the cutoff
    of some mobile non-alchemical atom (what a mobile-only pass must read), and the in-range pairs of those atoms."""
    from scipy.spatial import cKDTree
    x = np.mod(system.positions, system.box)
    alch = np.zeros(system.n_atoms, bool); alch[np.asarray(system.alchemical_atoms)] = True
    i_atoms = np.nonzero((system.mass > 0) & ~alch)[0]
    others = np.nonzero(~alch)[0]
    tree = cKDTree(x[others], boxsize=system.box)
    nb = tree.query_ball_point(x[i_atoms], system.cutoff)
    touched = set()
    pairs = 0
    for l in nb:
        touched.update(l); pairs += len(l) - 1    # (minus the atom itself; bonded exclusions are a handful per atom)
    return len(i_atoms), len(touched), pairs


def memory_use():
    """Peak resident host memory of this rank and the device memory in use on its GPU (GiB), for sizing chains per GPU."""
    import resource
    import torch
    free, total = torch.cuda.mem_get_info()
    out = {"host_peak_rss_gib": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0,
           "device_in_use_gib": (total - free) / 2.0 ** 30, "device_total_gib": total / 2.0 ** 30}
    try:   # the engines' device buffers end in 256 bytes nobody owns: a kernel that wrote past a buffer's end has left a mark there
        import ctypes
        from blues_amd import _lib
        g = (ctypes.c_int64 * 8)()
        if _lib.load().blues_debug_check_guards(g) == 0:
            out["device_buffer_guards"] = {"blocks": int(g[0]), "marked": int(g[1])}
    except Exception:
        pass
    return out


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment) BEFORE this process touches the GPU -- the parent never initialises HIP,
    never re-execs, only waits.  Rank 0 prints the one JSON line on the inherited stdout.  Returns non-zero if any rank
    failed (the others are then terminated: a dead rank would leave them waiting in a collective)."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BLUES_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, code))
                for q in pending:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def launch_check(args):
    """CPU-only check of the launch path (tests/test_bench_launcher.py): rendezvous, the per-iteration all-gather of the
    accept records and the max-over-ranks timing reduction on the gloo backend, no engine, no GPU."""
    import torch
    import torch.distributed as dist
    from blues_amd.replicas import env_rank, gather_decision_block, init_process_group
    rank, local_rank, world = env_rank()
    if world > 1:
        init_process_group("gloo")
    block = np.array([[1.0, 0.0, -0.5 * (rank + 1), 10.0 + rank, 0.0]] * 2)
    recs = gather_decision_block(block)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "records": int(recs.shape[0]), "work": recs[:, 3].tolist(), "max_rank_time": float(t.item())}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--replicas", type=int, default=2048, help="independent chains per GPU (a multiple of 256 per batch fills the 256 CUs evenly: the "
                    "nonbonded launch places one workgroup per chain).  The default, 2048 chains as two batches of 1024, is 48 GB of the 288 GB of "
                    "HBM and 4-5 s of set-up; one batch of 1024 advances a chain-step in 0.88 us through the driver, 0.95 at 512 (DESIGN.md section 4d)")
    ap.add_argument("--groups", type=int, default=2, help="the rank's chains form this many replica batches, each driven from its own host thread on its "
                    "own stream.  The batches' STEPPING calls take turns on the device (the other device-side calls of a batch -- energy prefetches, State "
                    "captures and restores, resets -- are short and may run beside the turn holder's kernels: the in-loop mean of a kernel moves by ~1 %%) while the "
                    "other batches' threads do their per-chain host work -- a tenth of an iteration's wall time with one batch (1 = a single batch); the rank's "
                    "all-gathers of the accept records follow, in iteration order, after the threads have finished")
    ap.add_argument("--concurrent", action="store_true", help="with --groups: no turns, the batches' kernels share the device (more ns/day; a kernel's "
                    "duration then includes its co-runners: DESIGN.md section 4d)")
    ap.add_argument("--workers", type=int, default=1, help="host threads for the per-chain plugin-boundary work inside a group")
    ap.add_argument("--nsteps-nc", type=int, default=NSTEPS_NC)
    ap.add_argument("--workload", default="rotmove", choices=["rotmove", "rotmove-solute", "water", "sidechain"],
                    help="rotmove = the benchmark (configs[1]: the ligand and the 87 nearest rigid waters mobile); rotmove-solute = the same switch with the mobile region of the "
                         "reference's freeze_radius (285 bonded solute atoms, all water frozen); water / sidechain = full-size runs of configs[3] / configs[4]")
    ap.add_argument("--md-steps", type=int, default=0, help="> 0: time the reference's FULL iteration (blues/simulation.py:1215-1257): every chain carries the md / alch / ncmc "
                    "triple, and each timed step is sync -> NCMC switch -> Metropolis -> reset -> this many MD steps on the unfrozen System.  `value` stays the "
                    "NCMC leg's ns/day (its share of the wall time); `full_iteration` has both legs.  Use --replicas 16..256 --groups 1 (an all-mobile engine is ~60 MB) and "
                    "--warmup 2: the SECOND iteration is the first whose NCMC leg receives a State from an MD leg, and the NCMC engines re-lay themselves out for it once")
    ap.add_argument("--no-alch", action="store_true", help="with --md-steps: no `alch` Simulation (the correction's energies then come from the NCMC engine at lambda = 1)")
    ap.add_argument("--decorrelate", type=int, default=250, help="set-up: steps of its own trajectory every chain runs (own velocities, own noise) before its hand-over State is "
                    "taken, so that the timed switches start from as many different states as there are chains (1 ps by default)")
    ap.add_argument("--setup-threads", type=int, default=None, help="host threads that create the chains (default: the cores this process may use divided among the ranks of the host, at most 16; 1 = one after the other)")
    ap.add_argument("--same-start", action="store_true", help="every chain starts every switch from the SAME coordinates and velocities (rounds 1-4)")
    ap.add_argument("--cpu-steps", type=int, default=12)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the single-chain measurement")
    ap.add_argument("--reciprocal", action="store_true", help="PME in full: reciprocal-space mesh, self, excluded-pair and dispersion terms on top of the direct-space sum "
                    "(the switching path north_star names is the direct-space one; this adds SURVEY.md 8f.2)")
    ap.add_argument("--launch-check", action="store_true", help="CPU-only check of the N-rank launch path (gloo, no engine)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend of the N-rank run: nccl (= RCCL over xGMI, the default) "
                    "or gloo (the accept records travel through host memory: what a one-GPU box can run)")
    ap.add_argument("--same-device", action="store_true", help="every rank uses GPU 0 (with --backend gloo: the whole N-rank path with real engines on a one-GPU box; "
                    "a correctness run -- the ranks share the device, `value` says nothing about scaling)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the stand-alone launches of the nonbonded kernel at the end (counter-collection runs: "
                    "every launch of the kernel in the trace is then one of the stepping loop)")
    args = ap.parse_args()

    # ---- N ranks: either a launcher (torch.distributed.run) already started us as one of them, or we start them ourselves
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if not args.launch_check:
            from blues_amd import build
            build.build_engine()      # hipcc only (no GPU call): compile once here instead of N times behind a lock
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if env_world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a line whose n_gpus is not what was asked for\n" % (args.gpus, env_world))
        sys.exit(2)
    if args.launch_check:
        sys.exit(launch_check(args))

    from blues_amd import build
    build.build_engine()
    import torch
    from blues_amd import simulation
    from blues_amd.replicas import env_rank, init_process_group, replica_seed
    rank, local_rank, world = env_rank()
    if args.same_device and args.backend == "nccl" and world > 1:
        sys.stderr.write("bench.py: --same-device needs --backend gloo (RCCL wants one device per rank)\n")
        sys.exit(2)
    device_index = 0 if args.same_device else local_rank
    if world > 1:
        if args.backend == "nccl":
            init_process_group("nccl")
        else:
            torch.cuda.set_device(device_index)
            init_process_group("gloo")
    import torch.distributed as dist

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.set_device(device_index)
    local_rank = device_index      # (what the chains are built on)
    nsteps, R = args.nsteps_nc, max(1, args.replicas)
    t_setup = time.perf_counter()
    from blues_amd import tuning
    G0 = max(1, min(args.groups, R))
    # the chains are laid out from the start as members of the batch they are about to join (BluesTuning.assume_batch: the layout a
    # batch of that size gives its members anyway), so that forming the batch re-lays nobody out: set-up time, nothing else
    from blues_amd.replicas import host_thread_share, share_host_threads
    host_threads = share_host_threads(world)   # (the ranks of one host share its cores: set-up threads here, re-sorts of several members inside the library)
    if args.setup_threads is None:
        args.setup_threads = host_threads
    with tuning.override(assume_batch=(R + G0 - 1) // G0):
        system, vel, chains = build_chains(rank, local_rank, nsteps, args.workload, R, reciprocal=args.reciprocal, md_steps=args.md_steps, with_alch=not args.no_alch, setup_threads=args.setup_threads)
    x0 = system.positions.copy()
    v0 = vel.copy()
    setup_parts = {"chains": time.perf_counter() - t_setup}
    _stage("chains built")

    # ---- configs[1] to the letter: ONE chain on the GPU, a lone engine with a lone engine's layout (its own construction, default tuning)
    single = None
    t_single = time.perf_counter()
    if rank == 0 and not args.no_single and not args.md_steps:
        clock = {"sync": 0.0, "switch": 0.0, "decide": 0.0}
        _, _, lone = build_chains(rank, local_rank, nsteps, args.workload, 1, reciprocal=args.reciprocal)
        st1 = md_states(lone, x0, v0)
        one_switch(None, lone, st1, nsteps, 0, clock)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(2):
            one_switch(None, lone, st1, nsteps, k, clock)
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t0) / 2
        e1 = lone[0]._ncmc_sim.context._engine
        k1_single = e1.time_nonbonded(50)
        a1 = ALGO_BYTES_PER_ATOM * system.n_atoms / (k1_single * 1e-6) / 1e9
        single = {"value": nsteps * DT_PS * 1e-3 / (dt1 / 86400.0), "unit": "ns/day", "ms_per_switch": 1e3 * dt1,
                  "roofline": {"bound": "latency", "achieved": a1, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a1 / HBM_PEAK_GBS,
                               "traffic": None, "usec_per_launch": k1_single,
                               "note": "one chain keeps a few of the 256 CUs busy: every kernel of its step is latency-bound (DESIGN.md section 4)"}}
        e1.close()
    t_single = time.perf_counter() - t_single      # (a measurement, not set-up: taken out of setup_seconds below)
    _stage("single replica measured")

    G = max(1, min(args.groups, R))
    bounds = [(g * R) // G for g in range(G + 1)]
    groups = [chains[bounds[g]:bounds[g + 1]] for g in range(G)]
    t_part = time.perf_counter()
    import threading
    turn = threading.Lock() if (G > 1 and not args.concurrent) else None
    # (isolate_failures: a chain that dies -- a move that blows its switch up -- is retired and counted in `chains_failed`; it does not end the other 2047)
    drivers = [simulation.BatchedBLUESSimulation(grp, workers=args.workers, device_turn=turn, isolate_failures=True) for grp in groups]
    setup_parts["batches"] = time.perf_counter() - t_part; t_part = time.perf_counter()
    if G > 1:   # chains driven from different threads draw from their own streams (reproducible whatever the interleaving)
        for c in chains:
            c._rng = np.random.RandomState(np.random.randint(0, 2 ** 31 - 1))
    full = args.md_steps > 0
    if full:
        # the full iteration: the chains' MD contexts own the state (reference simulation.py:1028-1037 copies it to the NCMC context at
        # the head of every iteration); a short MD leg per chain before the timed iterations gives every chain its own state
        if G != 1:
            sys.stderr.write("bench.py: --md-steps runs one replica batch per rank (--groups 1)\n"); sys.exit(2)
        from blues_amd import unit
        for c in chains:
            c._md_sim.context.setPositions(unit.Quantity(x0, "nanometer")); c._md_sim.context.setVelocities(unit.Quantity(v0, "nanometer/picosecond"))
        gstates = [None]
    else:
        gstates = [md_states(grp, x0, v0, batch=drv._ncmc_batch, driver=drv, decorrelate=0 if args.same_start else args.decorrelate) for grp, drv in zip(groups, drivers)]
    setup_parts["hand_over_states"] = time.perf_counter() - t_part
    _stage("batches made, states handed over")
    t_setup = time.perf_counter() - t_setup - t_single
    clocks = [{"sync": 0.0, "switch": 0.0, "decide": 0.0} for _ in range(G)]

    def switch_group(g, it):
        if full:
            return one_iteration(drivers[g], groups[g], nsteps, args.md_steps, it, clocks[g])
        return one_switch(drivers[g], groups[g], gstates[g], nsteps, it, clocks[g], gather=False)

    from blues_amd.replicas import gather_decision_block

    def switch_all(it):
        if full:
            return switch_group(0, it)
        return gather_decision_block(np.concatenate([switch_group(g, it) for g in range(G)]))

    for w in range(args.warmup):
        switch_all(w)      # on the main thread: every kernel variant has been launched once before threads start
    _stage("warm-up done")
    # what one batch's iteration takes when it has the device to itself (the warm-up ran the batches one after the other): the batches' threads
    # start a fraction of it apart, in the phase the turn-taking settles into anyway -- started together, the batch that loses the first turn
    # spends the first half switch of the other one waiting inside its first timed iteration (1.52 s against 1.36 s for every later one)
    alone = min((min(ck["iterations"]) for ck in clocks if ck.get("iterations")), default=0.0)
    engs = [c._ncmc_sim.context._engine for c in chains]
    st0 = engs[0].stats(); b0 = [d._ncmc_batch.stats() for d in drivers]
    for ck, drv in zip(clocks, drivers):
        ck.update({"sync": 0.0, "switch": 0.0, "decide": 0.0, "md": 0.0, "iterations": [], "layout_events": []})
        if hasattr(drv._ncmc_batch, "counters"):
            ck["_counters"] = drv._ncmc_batch.counters()
    # the nonbonded kernel is timed WHERE IT RUNS: every 4th force launch of the timed switches is bracketed by two HIP events on
    # the batch's stream (blues_batch_kernel_timing); that mean is roofline.usec_per_launch, what rocprofv3 averages for the same loop
    timing_batch = None if args.no_kernel_timing or not hasattr(drivers[0]._ncmc_batch, "kernel_timing") else drivers[0]._ncmc_batch
    if timing_batch is not None:
        timing_batch.kernel_timing(4)
    # the chains' Python objects (contexts, integrators, move engines, state tables: a few thousand per chain) live as long as the
    # process: park them in the permanent generation, so that the cyclic collector stops walking them during the iterations
    import gc
    gc.collect(); gc.freeze()
    barrier()
    t0 = time.perf_counter()
    recs = []
    if G == 1:
        for k in range(args.steps):
            recs.append(switch_all(k))
    elif turn is not None:
        # the batches take turns on the device: every thread runs its batch through all the iterations on its own (a batch steps
        # while the others' threads are in their host phases; nobody waits at an iteration boundary), and the rank's all-gathers
        # of the accept records -- bookkeeping, one per iteration -- follow in iteration order
        from concurrent.futures import ThreadPoolExecutor
        def run_batch(g):
            if g:
                time.sleep(0.22 * alone * g / max(1, G - 1))       # (inside the timed region; the batch would have spent it waiting for its first turn)
            return [switch_group(g, k) for k in range(args.steps)]
        with ThreadPoolExecutor(max_workers=G) as pool:
            parts = list(pool.map(run_batch, range(G)))
        for k in range(args.steps):
            recs.append(gather_decision_block(np.concatenate([parts[g][k] for g in range(G)])))
    else:
        # --concurrent: the groups run one iteration side by side on their own threads and streams: while they step, one group's
        # latency-bound kernels (list rebuilds, integrator) overlap another's compute-bound ones.  They are joined every
        # iteration (letting them drift apart was measured slower: a group's small per-chain operations then queue behind
        # the other groups' long launches), then the rank performs its one all-gather.
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=G) as pool:
            for k in range(args.steps):
                parts = list(pool.map(lambda g: switch_group(g, k), range(G)))
                recs.append(gather_decision_block(np.concatenate(parts)))
    barrier()
    elapsed = time.perf_counter() - t0
    _stage("timed iterations done")
    st1 = engs[0].stats(); b1 = [d._ncmc_batch.stats() for d in drivers]
    iteration_seconds = [list(ck.get("iterations", [])) for ck in clocks]     # per batch, per timed iteration (with several batches: the turns it waited for included)
    layout_events = [list(ck.get("layout_events", [])) for ck in clocks]       # per batch, per timed iteration
    layout_now = [d._ncmc_batch.counters() if hasattr(d._ncmc_batch, "counters") else {} for d in drivers]
    clock = {k: max(ck[k] for ck in clocks) for k in ("sync", "switch", "decide", "md") if k in clocks[0]}
    b0 = {k: sum(b[k] for b in b0) / G for k in b0[0]}; b1 = {k: sum(b[k] for b in b1) / G for k in b1[0]}
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if (world == 1 or args.backend == "nccl") else "cpu")
    tmin = t.clone()
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
    rank_elapsed = {"max": float(t.item()), "min": float(tmin.item()), "this_rank": elapsed}   # (host imbalance between ranks shows here first)
    from blues_amd.replicas import gather_rank_numbers
    per_rank = gather_rank_numbers([t_setup, memory_use()["host_peak_rss_gib"], float(os.getpid()), float(sum(len(d.dead) for d in drivers)),
                                    (clock["sync"] + clock["decide"]) / max(1, args.steps)])   # what N ranks on one host cost: set-up seconds, peak host memory; chains retired after a failure; the plugin boundary's seconds per iteration (hand-over + Metropolis + reset: per-chain host work)
    elapsed = float(t.item())

    # the kernel north_star prices against the HBM roofline: the direct-space nonbonded kernel, timed alone with HIP
    # events on the batch's own stream; one launch processes all R chains of this rank
    # (with pruned per-atom lists an atom is served from its current pruned list or -- a few percent of the atoms of a pass -- from its
    # full list while the pruned one is re-derived: both kinds of launch are timed, k1_us is their mean weighted with the share of
    # the second kind in THIS run's force passes; profiles/ holds the rocprofv3 average over the stepping loop beside it)
    k1_pruned = k1_full = k1_us = prune_share = k1_alone = None; k1_loop = None
    if timing_batch is not None:
        k1_loop = timing_batch.kernel_timing_result(); timing_batch.kernel_timing(0)
        if k1_loop["launches"] > 0:
            k1_us = k1_loop["usec"]
        # extras: the kernel alone (back-to-back launches, nothing else on the device), over current pruned lists and with every
        # atom re-deriving its list, and their mean weighted with the share of re-deriving atoms of this run
        k1_pruned, k1_full, prune_share = drivers[0]._ncmc_batch.time_nonbonded_modes(50)
        k1_alone = (1.0 - prune_share) * k1_pruned + prune_share * k1_full
        if k1_us is None:      # (a layout whose force kernel is not the per-atom-list one has no in-loop hook: the stand-alone figure, said so below)
            k1_us = k1_alone
    R_launch = len(groups[0])
    if rank == 0:
        n_atoms = system.n_atoms
        ms_per_step = 1e3 * elapsed / args.steps
        ns_day = world * R * args.steps * nsteps * DT_PS * 1e-3 / (elapsed / 86400.0)
        full_iteration = None
        if full:
            # `value` keeps BASELINE.json's definition (the switching leg: state sync + _stepNCMC + Metropolis + reset, MD leg excluded,
            # SURVEY.md 8d) -- here from the legs' clocks of the full iterations; both legs together are in full_iteration
            t_ncmc = (clock["sync"] + clock["switch"] + clock["decide"]) / args.steps
            t_md = clock["md"] / args.steps
            mde = chains[0]._md_sim.context._engine.stats()
            nce = chains[0]._ncmc_sim.context._engine.stats()   # (after MD legs the NCMC System's mobile atoms have scattered: its engine has moved to fragment lists)
            full_iteration = {"ns_day_both_legs": world * R * args.steps * (nsteps + args.md_steps) * DT_PS * 1e-3 / (elapsed / 86400.0),
                              "ns_day_md_leg": world * R * args.md_steps * DT_PS * 1e-3 / (t_md / 86400.0) if t_md > 0 else None,
                              "ms_sync": 1e3 * clock["sync"] / args.steps, "ms_ncmc": 1e3 * clock["switch"] / args.steps, "ms_boundary": 1e3 * clock["decide"] / args.steps,
                              "ms_md": 1e3 * t_md, "md_steps": args.md_steps, "ncmc_share_of_wall": t_ncmc / (t_ncmc + t_md),
                              "us_per_chain_step_md": 1e6 * t_md / (R * args.md_steps), "us_per_chain_step_ncmc": 1e6 * clock["switch"] / args.steps / (R * nsteps),
                              "md_engine": {"nonbonded_kernel": mde["nonbonded_kernel"], "list_builds": mde["list_builds"], "chain_prunes": mde["atom_prunes"], "force_passes": mde["force_passes"]},
                              "ncmc_engine": {"nonbonded_kernel": nce["nonbonded_kernel"], "resorts": nce["resorts"], "list_builds": nce["list_builds"]},
                              "triple": "md + alch + ncmc Simulations per chain" if chains[0]._alch_sim is not None else "md + ncmc Simulations per chain",
                              "reference": "blues/simulation.py:1215-1257 (run), 1189-1213 (_stepMD), 768-809 (the triple)"}
            ns_day = world * R * nsteps * DT_PS * 1e-3 / (t_ncmc / 86400.0)
            ms_per_step = 1e3 * elapsed / args.steps
        algo = ALGO_BYTES_PER_ATOM * n_atoms * R_launch
        last = np.asarray(recs[-1])
        # ---- roofline block of the dominant kernel.  north_star prices it against HBM with 36 B per atom per force evaluation
        # over ALL atoms; the kernel computes forces for the mobile atoms only, so the byte count of what such a pass has to touch
        # is given beside it, and so is the limit that actually binds (VALU issue), each from counters taken on this build.
        n_i, n_touched, n_pairs = in_range_counts(system)
        mob_bytes = (24.0 * n_touched + 12.0 * n_i) * R_launch
        ev = pmc_evidence(args.workload, R_launch)
        est = engs[0].stats()
        roofline = None
        if k1_us is not None:
          achieved = algo / (k1_us * 1e-6) / 1e9
          secs = k1_us * 1e-6
          in_loop = k1_loop is not None and k1_loop["launches"] > 0
          roofline = {"bound": "valu" if ev else "valu (counters not taken on this build: see profiles/README.md)",
                      "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                      "kernel": "%s (direct-space LJ + erfc Coulomb; one launch = %d chains; %s)" % ({0: "k_nonbonded_b", 1: "k_nonbonded_sub_b", 2: "k_nonbonded_atom_b", 3: "k_nonbonded_frag_b"}[est["nonbonded_kernel"]], R_launch,
                                 "HIP events around every 4th force launch of the timed switches, on the stream the kernel runs on" if in_loop else "timed alone with HIP events"),
                      "usec_per_launch": k1_us, "timed": "in the stepping loop" if in_loop else "alone",
                      "launches_timed": k1_loop["launches"] if in_loop else 50, "usec_longest_launch": k1_loop["usec_max"] if in_loop else None,
                      "usec_per_launch_alone": {"pruned_lists": k1_pruned, "re_deriving_every_list": k1_full, "share_of_atoms_re_deriving": prune_share, "weighted": k1_alone},
                      "algorithmic_bytes_per_launch": algo,
                      "algorithmic_bytes_definition": "36 B x all %d atoms x %d chains (SURVEY.md 8d: nominal, defined on all atoms)" % (n_atoms, R_launch),
                      "traffic": ev["traffic_bytes_per_launch"] if ev and "traffic_bytes_per_launch" in ev else None,
                      # the same roofline on MEASURED HBM bytes (north_star: "evidenced by rocprof HBM GB/s"): the counters' traffic over the launch's
                      # duration -- an upper bound where the reads are gathers (FETCH_SIZE is doubled: profiles/README.md); null without counters of this build
                      "achieved_counter": (ev["traffic_bytes_per_launch"] / secs / 1e9) if ev and "traffic_bytes_per_launch" in ev else None,
                      "frac_counter": (ev["traffic_bytes_per_launch"] / secs / 1e9 / HBM_PEAK_GBS) if ev and "traffic_bytes_per_launch" in ev else None,
                      "mobile_only": {"algorithmic_bytes": mob_bytes, "achieved": mob_bytes / secs / 1e9, "frac": mob_bytes / secs / 1e9 / HBM_PEAK_GBS,
                                      "definition": "24 B x %d environment atoms within the cutoff of a mobile atom + 12 B x %d mobile atoms, per chain" % (n_touched, n_i)},
                      "pairs": {"in_range_per_launch": n_pairs * R_launch,
                                "listed_per_launch": (est["pruned_list_entries"] if est.get("pruned_lists") else est["atom_list_entries"]) * R_launch,
                                "full_lists_per_launch": est["atom_list_entries"] * R_launch,
                                "lane_efficiency": n_pairs / max(1.0, 64.0 * (est["pruned_list_iterations"] if est.get("pruned_lists") else est["atom_list_iterations"])),
                                "note": "lane efficiency = pairs inside the cutoff / (wave iterations x 64 lanes) of the lists the kernel walks (the pruned per-atom lists; chain 0, end of the run)"}}
          if ev:
              c = ev["counters_per_launch"]
              insts = c.get("SQ_INSTS_VALU")
              if insts:
                  peak = 1024 * 2.4e9 / 2.0    # SIMDs x clock / 2 cycles per wave64 instruction (scripts/valu_issue.hip: 1.03 T/s reached with plain v_fma_f32)
                  v = {"insts_per_launch": insts, "issue_rate": insts / secs, "peak": peak, "frac": insts / secs / peak, "unit": "wave-instructions/s"}
                  # (SQ_ACTIVE_INST_VALU is NOT reported as busy time: on gfx950 it counts one quad-cycle per VALU instruction -- it equals
                  # SQ_INSTS_VALU to 2 % in every kernel measured)
                  if c.get("SQ_INSTS_VALU_TRANS_F32"):
                      v["transcendental_insts_per_launch"] = c["SQ_INSTS_VALU_TRANS_F32"]
                  roofline["valu"] = v
              roofline["pmc_source"] = {"file": PMC_FILE, "source_sha": ev["source_sha"], "kernel": ev["kernel"]}
        # chains that were retired after a failure (BatchedBLUESSimulation.isolate_failures: a switch that blew up) do not count
        chains_failed = int(per_rank[:, 3].sum())
        if chains_failed:
            ns_day *= (world * R - chains_failed) / float(world * R)
        out = {
            "metric": "NCMC ns/day (23k-atom toluene box, 1000-step switch, RandomLigandRotationMove), aggregate over independent chains",
            "value": ns_day, "unit": "ns/day", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 pair math / f64 accumulation, f64 alchemical+integrator", "data": "synthetic",
            "data_note": ("every chain carries its own state from its own MD legs (the warm-up iterations included one)" if full else
                          "every chain starts every switch from the same coordinates and velocities (its Philox stream differs): rebuild statistics are those of the first 4 ps from one geometry" if (args.same_start or args.decorrelate <= 0) else
                          "every chain starts its switches from its OWN state (%d steps of its own trajectory at set-up: own velocities, own noise, lambda <= %g throughout -- legs of %d steps with the integrator reset in between); there is no MD leg between the timed switches (--md-steps adds it)" % (args.decorrelate, DECORRELATE_LEG / float(nsteps), DECORRELATE_LEG)),
            "nonbonded_method": "PME direct space only" if system.nonbonded_method == 1 else "PME direct + reciprocal space (mesh %dx%dx%d, order %d), dispersion correction %s" % (tuple(system.pme_grid) + (system.pme_order, "on" if system.dispersion_correction else "off")),
            "config": {"workload": "S23k %s: %d atoms, %d mobile, %d alchemical, nstepsNC=%d, dt=4fs; %d independent chains per GPU in %d replica batch(es)"
                       % (args.workload, n_atoms, int((system.mass > 0).sum()), len(system.alchemical_atoms), nsteps, R, G),
                       "replicas_per_gpu": R, "batches_per_gpu": G, "host_workers": args.workers,
                       "parallelism": "%d replica batch(es) x %d chains per gpu%s, %d gpu(s)" % (G, R_launch, "" if G == 1 else (" taking turns on the device" if turn is not None else " sharing the device"), world),
                       "batches_take_turns": turn is not None},
            "chains_failed": chains_failed,
            "roofline": roofline,
            "full_iteration": full_iteration,
            "single_replica": single,
            "rank_elapsed_seconds": rank_elapsed,
            "per_rank": {"setup_seconds": per_rank[:, 0].tolist(), "host_peak_rss_gib": per_rank[:, 1].tolist(), "distinct_processes": int(len(set(per_rank[:, 2].tolist()))),
                         "boundary_seconds_per_iteration": per_rank[:, 4].tolist(),
                         "nproc": os.cpu_count(), "usable_cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None},
            "process_group": {"backend": (args.backend if world > 1 else None), "same_device": bool(args.same_device),
                              "replica_seeds_first_chain_of_each_rank": [int(replica_seed(1234, r * R)) for r in range(world)]},
            "memory": memory_use(),
            "engine": {"seconds": {k: v / args.steps for k, v in clock.items()}, "iteration_seconds_by_batch": iteration_seconds,
                       "iteration_seconds_max_over_median": max(max(it) / float(np.median(it)) for it in iteration_seconds if it) if any(iteration_seconds) else None,
                       # what the batches' layouts cost inside the timed iterations (blues_batch_get_counters): totals over the batches, then per
                       # batch and iteration wherever something happened -- a slow iteration names its cause
                       "replans": sum(e["replans"] for ev in layout_events for e in ev), "replan_seconds": sum(e["replan_seconds"] for ev in layout_events for e in ev),
                       "relayouts": sum(e["relayouts"] for ev in layout_events for e in ev), "reshapes": sum(e["reshapes"] for ev in layout_events for e in ev),
                       "resorts": sum(e["poll_resorts"] for ev in layout_events for e in ev), "resort_seconds": sum(e["resort_seconds"] for ev in layout_events for e in ev),
                       "straggled": sum(e["straggled"] for ev in layout_events for e in ev), "rejoined": sum(e["rejoined"] for ev in layout_events for e in ev),
                       "straggle_seconds": sum(e.get("straggle_seconds", 0.0) for ev in layout_events for e in ev),
                       "partial_steps": sum(e.get("partial_steps", 0) for ev in layout_events for e in ev),
                       "layout_events_by_batch": [{str(k): e for k, e in enumerate(ev) if any(e.values())} for ev in layout_events],
                       "layout_shape_by_batch": [{k: c.get(k) for k in ("tiles_per_list", "jcap", "nonbonded_kernel", "stragglers")} for c in layout_now],
                       "host_threads": host_threads,
                       "setup_seconds": t_setup, "setup_seconds_by_part": setup_parts, "setup_threads": args.setup_threads,
                       "plugin_boundary": ("one call per operation for all chains (blues_batch_*)" + ("" if drivers[0]._move_batchable() else "; the Move's hooks chain by chain")) if drivers[0]._batchable() else "chain by chain",
                       "force_passes_per_switch": (st1["force_passes"] - st0["force_passes"]) / args.steps,
                       "list_rebuilds_per_switch": (st1["list_generation"] - st0["list_generation"]) / args.steps,
                       "own_energy_evaluations_per_switch": (st1["own_energy_evaluations"] - st0["own_energy_evaluations"]) / args.steps,
                       "lockstep_steps_per_switch": (b1["lockstep_steps"] - b0["lockstep_steps"]) / args.steps,
                       "fallback_steps_per_switch": (b1["fallback_steps"] - b0["fallback_steps"]) / args.steps,
                       "i_tiles": st1["i_tiles"], "clusters": st1["clusters"], "jcap": st1["jcap"], "max_jcount": st1["max_jcount"]},
            "accept_records_last": {"chains": int(last.shape[0]), "accepted": int(last[:, 0].sum()),
                                    "mean_protocol_work_kj": float(np.nanmean(last[:, 3])), "first": last[0].tolist()},
        }
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(system, vel, args.cpu_steps)
        print(json.dumps(out), flush=True)   # (flushed before the chains are torn down at the end of main(): a fault there must not take the line with it; nothing is written after it, on either stream)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
