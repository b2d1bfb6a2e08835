# usage (GPU box): bash scripts/dev_single_timeline.sh [tuning] [first_step_shown] -- start/end (us, relative) of every kernel of three consecutive steps of a LONE chain (configs[1])
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/timeline1; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python3 scripts/dev_single.py "$1" > $out/log 2>&1
cat $out/log | tail -2
python3 - <<PY
import csv, glob
f = glob.glob("$out/t/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_step_default" in r[2]]
first = int("${2:-600}")
a, b = idx[first], idx[first + 3]
t0 = rows[a][1]
for r in rows[a + 1:b + 1]:
    print("  %8.1f -> %8.1f  (%6.1f)  %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2][:70]))
f = glob.glob("$out/t/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-60s calls %6s avg %9.2f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
rm -rf $out/t
