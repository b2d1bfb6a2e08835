#!/bin/bash
# usage (GPU box): bash scripts/dev_frag_prof.sh <tag> [R] [tuning spec]  -- dev_frag.py, then its stepping loop under rocprofv3 --kernel-trace --stats
tag=$1; R=${2:-16}; spec=${3:-}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/frag_$tag; rm -rf $out; mkdir -p $out
timeout 400 python3 scripts/dev_frag.py --R $R --nsteps 200 "$spec" > $out/dev.log 2>&1
tail -6 $out/dev.log
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 scripts/dev_frag.py --R $R --nsteps 200 --no-parity "$spec" > $out/stats.log 2>&1
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null; rm -rf $out/stats
python3 - <<PY
import csv, os
p = "$out/kernel_stats.csv"
if os.path.exists(p):
    for r in list(csv.DictReader(open(p)))[:16]:
        print("   %-66s calls %6s avg %9.2f us  %5.1f%%" % (r["Name"][:66], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
else:
    print(open("$out/stats.log").read()[-2000:])
PY
