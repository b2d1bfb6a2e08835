# usage (GPU box): bash scripts/dev_trace_hist.sh R nsteps tuning pattern -- per-call durations of the kernels matching `pattern` in the batched stepping path
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/trace_hist; rm -rf $out; mkdir -p $out
BLUES_TUNING=$3 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 scripts/batch_scaling.py --nsteps $2 $1 > $out/log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/t/*/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "$4" in n: d[n[:40]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for n, v in d.items():
    v.sort()
    print(n, len(v)); print("   ", " ".join("%.0f" % x[1] for x in v[:120]))
PY
rm -rf $out/t
