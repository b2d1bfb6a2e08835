"""CPU suite, part 3: the N>1 path -- one process per replica, gloo backend, world_size 2.
The data path has no collective; the only exchange is the all-gather of accept records."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_q):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import torch.distributed as dist
    from blues_amd import replicas
    r, lr, w = replicas.init_process_group("gloo")
    assert (r, w) == (rank, world)
    seed = replicas.replica_seed(1234, rank)
    recs = replicas.gather_decisions(accept=(rank == 1), iteration=7, log_accept=-1.5 * (rank + 1), protocol_work=10.0 + rank, correction=0.25)
    # several chains per rank (a replica batch): one (R, 5) block per rank, gathered rank-major
    block = np.array([[float(c == rank), 7.0, -0.5 * (3 * rank + c + 1), 20.0 + 3 * rank + c, 0.0] for c in range(3)])
    blk = replicas.gather_decision_block(block)
    assert blk.shape == (3 * world, 5) and blk[:, 3].tolist() == [20.0 + i for i in range(3 * world)]
    out_q.put((rank, seed, np.asarray(recs).tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_decisions_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60); assert p.exitcode == 0
    (r0, s0, rec0), (r1, s1, rec1) = res
    assert s0 != s1                                  # independent Philox keys per replica
    assert rec0 == rec1                              # every rank holds the same gathered table
    rec = np.array(rec0)
    assert rec.shape == (2, 5)
    assert rec[:, 0].tolist() == [0.0, 1.0] and rec[:, 1].tolist() == [7.0, 7.0]
    assert rec[:, 2].tolist() == [-1.5, -3.0] and rec[:, 3].tolist() == [10.0, 11.0]
    from blues_amd import replicas
    summ = replicas.acceptance_summary(rec)
    assert summ["replicas"] == 2 and summ["accepted"] == 1


def test_single_process_gather_needs_no_group():
    from blues_amd import replicas
    rec = replicas.gather_decisions(True, 0, -0.5, 3.0)
    assert rec.shape == (1, 5) and rec[0, 0] == 1.0
    assert replicas.replica_seed(1, 0) != replicas.replica_seed(1, 1) != replicas.replica_seed(2, 1)
    assert replicas.gather_decision_block(np.zeros((4, 5))).shape == (4, 5)
