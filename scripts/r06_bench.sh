# usage (GPU box): bash scripts/r06_bench.sh tag [steps] [warmup] -- the round-end bench command, line under gpurun_out/r06/, a one-line summary on stdout
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; ST=${2:-20}; WU=${3:-5}
mkdir -p gpurun_out/r06
timeout 900 python3 bench.py --gpus 1 --steps $ST --warmup $WU > gpurun_out/r06/bench_$TAG.json 2> gpurun_out/r06/bench_$TAG.err; echo "bench rc $?"
python3 - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06/bench_$TAG.json"))
    e = d["engine"]
    print("value %.1f k ns/day, ms/step %.1f, max/median %.3f, replans %s (%.3f s), relayouts %s, reshapes %s, resorts %s (%.3f s), K1 %.1f us frac %.3f, shape %s, setup %.1f s, rss %.1f GiB, failed %s" % (
        d["value"] / 1e3, d["ms_per_step"], e["iteration_seconds_max_over_median"], e["replans"], e["replan_seconds"], e["relayouts"], e["reshapes"], e["resorts"], e["resort_seconds"],
        d["roofline"]["usec_per_launch"], d["roofline"]["frac"], e["layout_shape_by_batch"], e["setup_seconds"], d["memory"]["host_peak_rss_gib"], d["chains_failed"]))
    print("events", e["layout_events_by_batch"], "single", d["single_replica"]["value"] if d.get("single_replica") else None)
except Exception as ex:
    print("no bench line:", ex); print(open("gpurun_out/r06/bench_$TAG.err").read()[-3000:])
PY
