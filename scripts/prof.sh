#!/bin/bash
# usage (on the GPU box): bash scripts/prof.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 1 --warmup 1 --no-cpu "$@" > gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_$tag/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print("%-52s calls %6s avg %9.2f us  %5.1f%%" % (r["Name"][:52], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
tail -1 gpurun_out/prof_$tag.log | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print('ns/day %.0f  ms/switch %.1f  K1 %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['usec_per_launch']))"
