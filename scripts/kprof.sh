#!/bin/bash
# usage (GPU box): bash scripts/kprof.sh <tag> [R] [nsteps] -- kernel stats of the batched stepping path alone (no driver), then SQ counters
tag=$1; R=${2:-256}; NS=${3:-200}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/kp_$tag; rm -rf $out; mkdir -p $out
python3 scripts/batch_scaling.py ${WORKLOAD:+--workload $WORKLOAD} --nsteps $NS $R > $out/plain.log 2>&1; cat $out/plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 scripts/batch_scaling.py ${WORKLOAD:+--workload $WORKLOAD} --nsteps $NS $R > $out/stats.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$out/stats/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-60s calls %6s avg %9.2f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
rm -f $out/stats/*/*kernel_trace.csv
if [ "$4" != "nosq" ]; then bash scripts/sq_counters.sh $tag $R | grep -A14 "nonbonded\|build_lists" ; fi
