"""Replica fan-out across the GPUs of one node.

The switching path shards by replica only: an NCMC switch is a serial chain of dependent
steps on ~1 MB of state, so each rank runs an independent BLUES chain (own seed, own MD
state) on its own GPU and there is NO data-path collective.  The one exchange per BLUES
iteration is an all-gather of a 32-byte decision record per rank -- RCCL over xGMI when the
tensors live on the GPU (backend "nccl"), gloo on CPU for tests.
"""
import os

import numpy as np

RECORD_FIELDS = ("accept", "iteration", "log_accept", "protocol_work", "correction")


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None):
    """One process per GPU; rendezvous from MASTER_ADDR/MASTER_PORT (127.0.0.1 on one node)."""
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if world <= 1 or dist.is_initialized():
        return rank, local_rank, world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    if backend is None:
        import torch
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        import torch
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def host_thread_share(world=None, cap=16):
    """Host threads a rank may use for work its chains share (set-up on host threads, re-sorts of several batch members at a poll): the
    cores this process may run on, divided among the ranks of the host, at most `cap`.  N ranks on one node must not each start 16
    threads on a 16-core quota (one process per GPU, host-side work included)."""
    if world is None:
        world = env_rank()[2]
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    # (local ranks: the launcher exports LOCAL_WORLD_SIZE; without it every rank is assumed to be on this host)
    local = int(os.environ.get("LOCAL_WORLD_SIZE", world) or world)
    return max(1, min(int(cap), cores // max(1, local)))


def share_host_threads(world=None):
    """Tells the native library this rank's share of the host's cores (BluesTuning.host_threads); returns it."""
    from . import tuning
    n = host_thread_share(world)
    tuning.set(host_threads=n)
    return n


def replica_seed(base_seed, rank):
    """Distinct, reproducible Philox keys per replica."""
    return (int(base_seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


def build_in_parallel(make, count, workers=None):
    """[make(0), ..., make(count - 1)] with the calls spread over host threads.  Creating a chain is mostly native host work (the
    engine derives exclusion tables, constraint clusters, fragments and the first sorted layout: 6-8 ms per 23k-atom chain; the ctypes
    call releases the interpreter lock), so the 2048 chains of a GPU are ready in a fraction of the serial time.  Order is kept;
    the first exception is re-raised.  workers: default this rank's share of the host's cores (host_thread_share); 1 = the plain loop."""
    if workers is None:
        workers = host_thread_share()
    if workers <= 1 or count <= 1:
        return [make(i) for i in range(count)]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(workers, count)) as pool:
        return list(pool.map(make, range(count)))


def gather_decisions(accept, iteration, log_accept, protocol_work, correction=0.0, device=None):
    """All-gather of {accept, iter, log_accept, protocol_work, correction} -> (world, 5) float64 array on every rank."""
    import torch
    import torch.distributed as dist
    rec = torch.tensor([float(bool(accept)), float(iteration), float(log_accept), float(protocol_work), float(correction)], dtype=torch.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec.numpy().reshape(1, -1)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    rec = rec.to(device)
    out = [torch.empty_like(rec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rec)
    return torch.stack(out).cpu().numpy()


def gather_decision_block(records, device=None):
    """The same exchange for a rank that runs several chains in one replica batch: `records` is (R, 5) (RECORD_FIELDS per
    chain); every rank receives the (world * R, 5) array, rank-major.  Still one collective per BLUES iteration."""
    import torch
    import torch.distributed as dist
    rec = torch.as_tensor(np.asarray(records, dtype=np.float64).reshape(-1, len(RECORD_FIELDS)))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec.numpy()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    rec = rec.to(device)
    out = [torch.empty_like(rec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rec)
    return torch.cat(out).cpu().numpy()


def gather_rank_numbers(values, device=None):
    """Per-rank diagnostics (set-up seconds, host memory, ...) -> (world, len(values)) on every rank; the same collective path as
    the accept records (bookkeeping: once per run)."""
    import torch
    import torch.distributed as dist
    rec = torch.as_tensor(np.asarray(values, dtype=np.float64).reshape(-1))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec.numpy().reshape(1, -1)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    rec = rec.to(device)
    out = [torch.empty_like(rec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rec)
    return torch.stack(out).cpu().numpy()


def acceptance_summary(records):
    r = np.asarray(records, dtype=np.float64).reshape(-1, len(RECORD_FIELDS))
    return {"replicas": int(r.shape[0]), "accepted": int(r[:, 0].sum()), "mean_log_accept": float(np.nanmean(r[:, 2])),
            "mean_work_kj": float(np.nanmean(r[:, 3]))}
