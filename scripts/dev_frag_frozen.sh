#!/bin/bash
# usage (GPU box): bash scripts/dev_frag_frozen.sh <tag> [R]  -- the headline (frozen) configuration through the per-atom lists (k1_mode 2) and through the fragment lists (k1_mode 3):
# golden-vector parity of both, stepping rate, kernel table
tag=$1; R=${2:-1024}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/frozen_$tag; rm -rf $out; mkdir -p $out
for mode in 3 2; do
  BLUES_TUNING=k1_mode=$mode timeout 600 python3 scripts/batch_scaling.py --nsteps 300 $R > $out/scaling_m$mode.log 2>&1; tail -2 $out/scaling_m$mode.log
  BLUES_TUNING=k1_mode=$mode timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 scripts/batch_scaling.py --nsteps 300 $R > $out/stats_m$mode.log 2>&1
  cp $out/stats/*/*kernel_stats.csv $out/kernel_stats_m$mode.csv 2>/dev/null; rm -rf $out/stats
  python3 - <<PY
import csv, os
p = "$out/kernel_stats_m$mode.csv"
if os.path.exists(p):
    for r in list(csv.DictReader(open(p)))[:12]:
        print("   %-66s calls %6s avg %9.2f us  %5.1f%%" % (r["Name"][:66], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
BLUES_TUNING=k1_mode=3 timeout 900 python3 -m pytest tests/test_gpu_s23k_golden.py -x -q 2>&1 | tail -15
