"""CPU suite: `python bench.py --gpus N` must start N ranks (the flag used to be parsed and ignored).
The launch path -- spawn before any GPU call, rendezvous on 127.0.0.1, the all-gather of accept records, the
max-over-ranks reduction, one JSON line from rank 0, non-zero exit when a rank dies -- is exercised here on the gloo
backend through bench.py's own --launch-check mode (no engine, no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_gpus_flag_spawns_ranks():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                      # one JSON line, from rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["records"] == 4 and out["work"] == [10.0, 10.0, 11.0, 11.0] and out["max_rank_time"] == 2.0


def test_launcher_started_ranks_are_respected_and_mismatch_is_refused():
    # started by a launcher as 1 of 1: no spawn
    r = _run(["--gpus", "1", "--launch-check"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    # --gpus disagrees with the launcher's world size: refuse instead of mislabelling n_gpus
    r = _run(["--gpus", "4", "--launch-check"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "refusing" in r.stderr


def test_eight_ranks_under_the_driver_launcher():
    """The round-end scaling run: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py --gpus 8`.  Eight ranks
    rendezvous on 127.0.0.1, every rank contributes its block of accept records to the all-gather in rank order, the timing
    reduction takes the slowest rank, rank 0 alone prints."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "8", "--launch-check"], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["records"] == 16 and out["max_rank_time"] == 8.0
    assert out["work"] == [10.0 + q // 2 for q in range(16)]       # two records per rank, in rank order


def test_help_text_renders():
    """argparse expands % in help strings: a literal percent sign must be doubled (round 5: `python bench.py --help` raised ValueError)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    assert "--md-steps" in out.stdout and "--setup-threads" in out.stdout
