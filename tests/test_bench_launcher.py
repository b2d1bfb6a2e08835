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
