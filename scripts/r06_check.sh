# usage (GPU box): bash scripts/r06_check.sh [tag] -- the GPU test suite, then the round-end bench command; everything under gpurun_out/r06/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-check}
mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r06/pytest_$TAG.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r06/pytest_$TAG.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_$TAG.json 2> gpurun_out/r06/bench_$TAG.err; echo "bench rc $?"
python3 - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06/bench_$TAG.json"))
    e = d["engine"]
    print("value %.1f k ns/day, ms/step %.1f, max/median %.3f, replans %s (%.3f s), relayouts %s, reshapes %s, resorts %s (%.3f s), K1 %.1f us frac %.3f, shape %s, setup %.1f s, rss %.1f GiB" % (
        d["value"] / 1e3, d["ms_per_step"], e["iteration_seconds_max_over_median"], e["replans"], e["replan_seconds"], e["relayouts"], e["reshapes"], e["resorts"], e["resort_seconds"],
        d["roofline"]["usec_per_launch"], d["roofline"]["frac"], e["layout_shape_by_batch"], e["setup_seconds"], d["memory"]["host_peak_rss_gib"]))
    print("events", e["layout_events_by_batch"])
except Exception as ex:
    print("no bench line:", ex); print(open("gpurun_out/r06/bench_$TAG.err").read()[-3000:])
PY
