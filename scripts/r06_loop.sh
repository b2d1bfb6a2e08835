# usage (GPU box): bash scripts/r06_loop.sh [n] -- the round-end bench command n times in a row on one box, stopping at the first run that leaves no line (its stderr is kept)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
N=${1:-8}
for i in $(seq 1 $N); do
  t0=$(date +%s.%N); timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/loop_$i.json 2> gpurun_out/r06/loop_$i.err; rc=$?; echo "   wall $(python3 -c "import time; print(round(time.time() - $t0, 1))") s, stages: $(grep -c "^\[bench\]" gpurun_out/r06/loop_$i.err)"
  python3 - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06/loop_$i.json")); e = d["engine"]
    print("run $i rc $rc: %.1f k ns/day, max/median %.3f, resorts %s straggled %s partial %s relayouts %s guards %s failed %s" % (d["value"] / 1e3, e["iteration_seconds_max_over_median"], e["resorts"], e["straggled"], e.get("partial_steps"), e["relayouts"], d["memory"].get("device_buffer_guards"), d["chains_failed"]))
except Exception as ex:
    print("run $i rc $rc: NO LINE (%s)" % ex); print(open("gpurun_out/r06/loop_$i.err").read()[-1500:])
PY
  [ -s gpurun_out/r06/loop_$i.json ] || break
done
