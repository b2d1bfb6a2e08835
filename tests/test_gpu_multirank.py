"""GPU suite, multi-rank part (needs >= 2 GPUs; skipped on the 1-GPU box): the one exchange of the path -- the all-gather
of accept records -- over RCCL (backend "nccl"), and bench.py's own N-rank launch."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    import torch
    return torch.cuda.device_count()   # counting devices does not initialise the GPU in this process


_WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from blues_amd import replicas
rank, local_rank, world = replicas.init_process_group("nccl")
assert dist.get_backend() == "nccl"
block = np.array([[float(c == rank), 3.0, -0.5 * (2 * rank + c + 1), 20.0 + 2 * rank + c, 0.0] for c in range(2)])
out = replicas.gather_decision_block(block)
assert out.shape == (2 * world, 5) and out[:, 3].tolist() == [20.0 + i for i in range(2 * world)], out
one = replicas.gather_decisions(rank == 1, 3, -1.0 - rank, 5.0 + rank)
assert one.shape == (world, 5) and one[:, 3].tolist() == [5.0 + r for r in range(world)]
dist.barrier(); torch.cuda.synchronize()
if rank == 0:
    print(json.dumps({"ok": True, "world": world}))
dist.destroy_process_group()
"""


def test_gather_decision_block_over_rccl():
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs")
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _WORKER % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    assert json.loads(outs[0][0].strip().splitlines()[-1]) == {"ok": True, "world": 2}


def test_bench_gpus_2_reports_two_ranks():
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--replicas", "8",
                        "--nsteps-nc", "200", "--no-cpu", "--no-single"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["accept_records_last"]["chains"] == 16 and np.isfinite(out["value"])


def test_two_ranks_with_real_engines_on_one_gpu():
    """The whole N-rank path short of RCCL itself, on the one-GPU box: bench.py starts two ranks, BOTH on device 0, each with its own
    replica batch of 8 real engines; the accept records travel over gloo.  16 records in rank order, distinct Philox keys per
    rank, both ranks' clocks in the line (the path shards by replica only, one exchange per iteration)."""
    if _n_gpus() < 1:
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "1", "--warmup", "0",
                        "--replicas", "8", "--nsteps-nc", "200", "--no-cpu", "--no-single"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["accept_records_last"]["chains"] == 16 and np.isfinite(out["value"])
    pg = out["process_group"]
    assert pg["backend"] == "gloo" and pg["same_device"] is True
    assert len(set(pg["replica_seeds_first_chain_of_each_rank"])) == 2
    el = out["rank_elapsed_seconds"]
    assert 0.0 < el["min"] <= el["max"] and el["max"] < 600.0
    # every chain of both ranks stepped the whole (short) protocol in lock step with its batch
    assert out["engine"]["lockstep_steps_per_switch"] >= 200 and out["engine"]["fallback_steps_per_switch"] == 0


def test_eight_ranks_with_real_engines_on_one_gpu():
    """Dress rehearsal of the 8-GPU run on the one-GPU box (BASELINE.json configs[2]): bench.py starts EIGHT ranks, all on
    device 0, each with its own replica batch of 8 real engines; the accept records travel over gloo.  Everything but RCCL and xGMI:
    eight processes building engines at once (one build lock, one rendezvous port), 64 records in rank order, eight distinct Philox
    keys, every rank's set-up time and host memory in the line."""
    if _n_gpus() < 1:
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--same-device", "--steps", "1", "--warmup", "0",
                        "--replicas", "8", "--groups", "1", "--nsteps-nc", "200", "--no-cpu", "--no-single"], env=env, capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["accept_records_last"]["chains"] == 64 and np.isfinite(out["value"])
    pg = out["process_group"]
    assert pg["backend"] == "gloo" and pg["same_device"] is True
    assert len(set(pg["replica_seeds_first_chain_of_each_rank"])) == 8
    pr = out["per_rank"]
    assert len(pr["setup_seconds"]) == 8 and pr["distinct_processes"] == 8 and all(0.0 < t < 600.0 for t in pr["setup_seconds"])
    assert len(pr["host_peak_rss_gib"]) == 8 and all(0.05 < m < 32.0 for m in pr["host_peak_rss_gib"]) and pr["nproc"] >= 1
    el = out["rank_elapsed_seconds"]
    assert 0.0 < el["min"] <= el["max"] and el["max"] < 900.0
    assert out["engine"]["lockstep_steps_per_switch"] >= 200 and out["engine"]["fallback_steps_per_switch"] == 0
    # eight ranks on one host share its cores: set-up threads and the library's re-sort pool are cores / ranks each (replicas.host_thread_share)
    import os as _os
    cores = len(_os.sched_getaffinity(0))
    assert out["engine"]["host_threads"] == max(1, min(16, cores // 8)) and out["engine"]["setup_threads"] == out["engine"]["host_threads"]
    # ... and the per-chain host work of the plugin boundary (hand-over, Metropolis test, reset) of a rank must not suffer from its seven
    # neighbours: against ONE rank running the same thing alone.  (All eight ranks share ONE GPU here, which the eight-GPU node does not:
    # the bound leaves room for their launches queueing behind each other.)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                         "--replicas", "8", "--groups", "1", "--nsteps-nc", "200", "--no-cpu", "--no-single"], env=env, capture_output=True, text=True, timeout=1200)
    assert r1.returncode == 0, r1.stderr[-3000:]
    alone = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1])["per_rank"]["boundary_seconds_per_iteration"][0]
    together = pr["boundary_seconds_per_iteration"]
    assert len(together) == 8 and max(together) <= 1.5 * alone + 0.05, (together, alone)
