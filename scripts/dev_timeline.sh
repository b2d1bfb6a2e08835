# usage (GPU box): bash scripts/dev_timeline.sh R nsteps tuning [first_step_shown] -- start/end (us, relative) of every kernel of a few consecutive steps of the batched stepping path
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/timeline; rm -rf $out; mkdir -p $out
BLUES_TUNING=$3 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 scripts/batch_scaling.py --nsteps $2 $1 > $out/log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$out/t/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(f))]
rows.sort()
# steps are delimited by k_step_default_b
idx = [i for i, r in enumerate(rows) if "k_step_default" in r[2]]
first = int("${4:-60}")
a, b = idx[first], idx[first + 3]
t0 = rows[a][1]
print("[$3]")
for r in rows[a + 1:b + 1]:
    print("  %8.1f -> %8.1f  (%6.1f)  q%-4s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[2][:48]))
PY
rm -rf $out/t
