# usage (GPU box): bash scripts/r06_step.sh tag [BLUES_TUNING] -- bare stepping of 1024 chains and the kernel table of the same loop (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; T=$2
mkdir -p gpurun_out/r06
out=gpurun_out/r06/step_$TAG; rm -rf $out; mkdir -p $out
BLUES_TUNING=$T timeout 600 python3 scripts/batch_scaling.py --nsteps 600 1024 > $out/plain.log 2>&1; grep 'us/step' $out/plain.log | tail -1
BLUES_TUNING=$T timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 scripts/batch_scaling.py --nsteps 600 1024 > $out/log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$out/stats/*/*kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if not r["Name"].startswith("__amd") and float(r["TotalDurationNs"]) > 3e5]
open("$out/kernel_stats.csv", "w").write(open(f).read())
for r in rows:
    print("   %-62s calls %6s avg %9.2f us tot %8.1f ms" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $out/stats
