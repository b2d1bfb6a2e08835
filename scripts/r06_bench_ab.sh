# usage (GPU box): bash scripts/r06_bench_ab.sh "<BLUES_TUNING A>" "<BLUES_TUNING B>" [steps] -- the default bench line under two tunings on ONE box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
ST=${3:-8}
i=0
for t in "$1" "$2"; do
  i=$((i+1))
  BLUES_TUNING="$t" timeout 900 python3 bench.py --gpus 1 --steps $ST --warmup 3 --no-cpu --no-single > gpurun_out/r06/bench_ab$i.json 2> gpurun_out/r06/bench_ab$i.err
  python3 - <<PY
import json
d = json.load(open("gpurun_out/r06/bench_ab$i.json")); e = d["engine"]
print("[%s] %.1f k ns/day, %.1f ms/iteration, K1 %.1f us frac %.3f, max/median %.3f, straggled %s, resorts %s, jcap %s" % ("$t", d["value"] / 1e3, d["ms_per_step"], d["roofline"]["usec_per_launch"], d["roofline"]["frac"], e["iteration_seconds_max_over_median"], e.get("straggled"), e.get("resorts"), [s["jcap"] for s in e["layout_shape_by_batch"]]))
PY
done
