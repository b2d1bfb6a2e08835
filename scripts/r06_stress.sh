# usage (GPU box): bash scripts/r06_stress.sh [n] [tuning] -- the default bench with a wider outer list margin (members leave their batch's layout, come back, are re-sorted at polls) n times: does it survive, what did the iterations cost
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
N=${1:-2}; T=${2:-skin=0.25}
for i in $(seq 1 $N); do
  BLUES_TUNING=$T timeout 800 python3 bench.py --gpus 1 --steps 24 --warmup 3 --no-cpu --no-single > gpurun_out/r06/stress_$i.json 2> gpurun_out/r06/stress_$i.err; rc=$?
  python3 - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06/stress_$i.json")); e = d["engine"]
    print("run $i rc $rc: %.1f k ns/day, max/median %.3f, resorts %s straggled %s rejoined %s partial %s fallback/switch %s failed %s" % (d["value"] / 1e3, e["iteration_seconds_max_over_median"], e["resorts"], e["straggled"], e["rejoined"], e.get("partial_steps"), e["fallback_steps_per_switch"], d["chains_failed"]))
except Exception as ex:
    print("run $i rc $rc: no line (%s)" % ex); print(open("gpurun_out/r06/stress_$i.err").read()[-600:])
PY
done
