#!/usr/bin/env python3
"""bench.py -- NCMC switching-leg throughput of the MI355X-native engine.

A "step" is one pass of the hot path over one batch of synthetic input: one complete
1000-step NCMC switch (BASELINE.json configs[1]) of the S23k box (23,400 atoms, 15-atom
alchemical toluene, ~276 mobile atoms emulating freeze_radius 5 A) driven through the
drop-in boundary exactly as BLUESSimulation drives it: state sync, _stepNCMC with the
RandomLigandRotationMove at lambda = 0.5, accept/reject, reset.  metric = ns/day of
switching trajectory, whole job (all ranks); every rank runs an independent replica
(weak scaling, no data-path collective; one all-gather of accept records per switch).

    python bench.py --gpus N --steps K --warmup W
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


def build_replica(rank, local_rank, nsteps, workload):
    from blues_amd import integrators, moves, simulation, systems
    from blues_amd.context import Simulation
    from blues_amd.replicas import replica_seed
    if workload == "water":  # configs[3]: nothing frozen
        system, vel = systems.s23k(frozen=False)
    else:
        system, vel = systems.s23k(mobile_atoms=275, frozen=True)
    integ = integrators.generateNCMCIntegrator(nstepsNC=nsteps, dt=DT_PS, temperature=300.0, seed=replica_seed(1234, rank))
    sim = Simulation(None, system, integ, device=local_rank, precision="mixed", replica=rank)
    lig = np.asarray(system.alchemical_atoms)
    mover = moves.MoveEngine(moves.RandomLigandRotationMove(lig, system.mass[lig], random_state=1000 + rank))
    blues = simulation.BLUESSimulation(simulation.SimulationSet(sim), {"nstepsNC": nsteps, "moveStep": nsteps // 2, "nIter": 1}, mover)
    sim.context.setVelocities(vel)
    return system, vel, sim, blues


def one_switch(blues, sim, x0, v0, nsteps, it):
    from blues_amd.replicas import gather_decisions
    sim.context.setPositions(x0)
    sim.context.setVelocities(v0)
    blues.currentIter = it
    blues._syncStatesMDtoNCMC()
    blues._stepNCMC(nsteps, nsteps // 2)
    blues._acceptRejectMove()
    rec = gather_decisions(blues.last["accept"], it, blues.last["log_accept"], blues.last["protocol_work"], blues.last["correction"])
    blues._resetSimulations(300.0)
    return rec


def cpu_baseline(system, vel, nsteps_sample):
    """The CPU oracle (fp64 restatement, single thread) on a bounded sample of the same workload."""
    from blues_amd import integrators
    from oracle import oracle
    integ = integrators.generateNCMCIntegrator(nstepsNC=NSTEPS_NC, dt=DT_PS, temperature=300.0, seed=1234)
    o = oracle.Oracle(system, integ.to_data())
    o.set_velocities(vel)
    o.step(1)  # first-step block + warm caches
    t0 = time.perf_counter()
    o.step(nsteps_sample)
    dt = time.perf_counter() - t0
    ns_day = nsteps_sample * DT_PS * 1e-3 / (dt / 86400.0)
    return {"value": ns_day, "unit": "ns/day", "cores": 1, "kind": "port",
            "sample": "%d NCMC steps of the same S23k switch (3 full fp64 energy/force evaluations per step), %.1f s" % (nsteps_sample, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nsteps-nc", type=int, default=NSTEPS_NC)
    ap.add_argument("--workload", default="rotmove", choices=["rotmove", "water"])
    ap.add_argument("--cpu-steps", type=int, default=12)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    from blues_amd import build
    build.build_engine()
    import torch
    from blues_amd.replicas import env_rank, init_process_group
    rank, local_rank, world = env_rank()
    if world > 1:
        init_process_group("nccl")
    import torch.distributed as dist

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.set_device(local_rank)
    nsteps = args.nsteps_nc
    system, vel, sim, blues = build_replica(rank, local_rank, nsteps, args.workload)
    x0 = system.positions.copy()
    v0 = vel.copy()
    for w in range(args.warmup):
        one_switch(blues, sim, x0, v0, nsteps, w)
    eng = sim.context._engine
    st0 = eng.stats()
    barrier()
    t0 = time.perf_counter()
    recs = []
    for k in range(args.steps):
        recs.append(one_switch(blues, sim, x0, v0, nsteps, k))
    barrier()
    elapsed = time.perf_counter() - t0
    st1 = eng.stats()
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # the kernel north_star prices against the HBM roofline: the direct-space nonbonded kernel, timed alone with HIP
    # events on the engine's own stream (blues_time_nonbonded)
    k1_us = eng.time_nonbonded(50)
    traffic = None
    try:  # HBM-side bytes per launch from separate rocprofv3 --pmc passes (profiles/README.md says how they were taken)
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
            pmc = json.load(fh).get(args.workload)
        if pmc:
            traffic = pmc["traffic_bytes_per_launch"]
    except Exception:
        traffic = None
    if rank == 0:
        n_atoms = system.n_atoms
        ms_per_step = 1e3 * elapsed / args.steps
        ns_day = world * args.steps * nsteps * DT_PS * 1e-3 / (elapsed / 86400.0)
        achieved = ALGO_BYTES_PER_ATOM * n_atoms / (k1_us * 1e-6) / 1e9
        out = {
            "metric": "NCMC ns/day (23k-atom toluene box, 1000-step switch, RandomLigandRotationMove)",
            "value": ns_day, "unit": "ns/day", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 pair math / f64 accumulation, f64 alchemical+integrator", "data": "synthetic",
            "config": {"workload": "S23k %s: %d atoms, %d mobile, 15 alchemical, nstepsNC=%d, dt=4fs, 1 replica per GPU"
                       % (args.workload, n_atoms, int((system.mass > 0).sum()), nsteps),
                       "parallelism": "replica-per-gpu x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_nonbonded (direct-space LJ + erfc Coulomb)", "usec_per_launch": k1_us,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ATOM * n_atoms},
            "engine": {"force_passes_per_switch": (st1["force_passes"] - st0["force_passes"]) / args.steps,
                       "kernel_launches_per_switch": (st1["kernel_launches"] - st0["kernel_launches"]) / args.steps,
                       "list_rebuilds_per_switch": (st1["list_generation"] - st0["list_generation"]) / args.steps,
                       "i_tiles": st1["i_tiles"], "clusters": st1["clusters"], "jcap": st1["jcap"], "npart": st1["npart"], "seg_len": st1["seg_len"], "wpb": st1["wpb"]},
            "accept_records_last": np.asarray(recs[-1]).tolist(),
        }
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(system, vel, args.cpu_steps)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
