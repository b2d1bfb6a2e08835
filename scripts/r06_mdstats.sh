# usage (GPU box): bash scripts/r06_mdstats.sh [R] -- kernel table of the FULL iteration (MD leg included) at R chains
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-64}
out=gpurun_out/r06/md_$R; rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --md-steps 1000 --replicas $R --groups 1 --steps 1 --warmup 1 --no-cpu --no-single > $out/log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$out/stats/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
open("$out/kernel_stats.csv", "w").write(open(f).read())
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("   %-70s calls %6s avg %9.2f us tot %8.1f ms %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -1 $out/log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['full_iteration'])"
rm -rf $out/stats
