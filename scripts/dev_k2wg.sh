# usage (GPU box): bash scripts/dev_k2wg.sh "24 11 8 5" [lib] -- the alchemical kernel's time against environment workgroups per chain (R = 512, no fork)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -n "$2" ]; then export BLUES_LIB_PATH=$PWD/$2; fi
for w in $1; do
  out=gpurun_out/k2wg_$w; rm -rf $out; mkdir -p $out
  BLUES_TUNING=fork=0,k2_workgroups=$w rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 scripts/batch_scaling.py --nsteps 120 512 > $out/log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$out/stats/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f))):
    if "alchemical" in r["Name"]: print("$2 wg=$w %-50s calls %6s avg %9.2f us  %5.1f%%" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  grep -i "us/step\|ns/day" $out/log | tail -1
  rm -f $out/stats/*/*kernel_trace.csv
done
