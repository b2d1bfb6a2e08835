# usage (GPU box): bash scripts/dev_k2dexp.sh -- floor experiments of the dense alchemical kernel (1: no pair arithmetic, 2: staging only, 3: no pair pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for e in 0 1 2 3; do
  if [ $e = 0 ]; then L=blues_amd/csrc/libblues_hip.so; else L=scripts/devtests/libk2d$e.so; fi
  out=gpurun_out/k2d_$e; rm -rf $out; mkdir -p $out
  BLUES_TUNING=fork=0 BLUES_LIB_PATH=$PWD/$L rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 scripts/batch_scaling.py --nsteps 100 512 > $out/log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$out/stats/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "dense" in r["Name"]: print("exp $e  %-50s calls %5s avg %8.2f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $out/stats
done
