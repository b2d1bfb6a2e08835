# usage (GPU box): bash scripts/dev_tunesweep.sh R nsteps "tuning1" "tuning2" ... -- kernel stats of the batched stepping path under BLUES_TUNING settings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$1; NS=$2; shift 2
i=0
for t in "$@"; do
  i=$((i+1)); out=gpurun_out/ts_$i; rm -rf $out; mkdir -p $out
  BLUES_TUNING=$t python3 scripts/batch_scaling.py --nsteps $NS $R > $out/plain.log 2>&1; echo "[$t] plain: $(grep 'us/step' $out/plain.log | tail -1)"
  BLUES_TUNING=$t rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 scripts/batch_scaling.py --nsteps $NS $R > $out/log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$out/stats/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f))):
    if r["Name"].startswith("__amd") or float(r["TotalDurationNs"]) < 3e5: continue
    print("   %-62s calls %6s avg %9.2f us tot %8.1f ms" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  rm -f $out/stats/*/*kernel_trace.csv
done
