#!/bin/bash
# usage (on the GPU box): bash scripts/evidence.sh <tag> [bench args...]
# bench line, rocprofv3 kernel stats of the same workload, and the two PMC passes for HBM traffic of the nonbonded kernel.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ev_$tag; rm -rf $out; mkdir -p $out
python3 bench.py "$@" > $out/bench.log 2>&1; tail -1 $out/bench.log > $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-single "$@" > $out/stats.log 2>&1; rm -f $out/stats/*/*kernel_trace.csv
for c in ${PMC_COUNTERS:-}; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-single --nsteps-nc 40 "$@" > $out/pmc_$c.log 2>&1
done
python3 - <<PY
import csv, glob, json
out = "$out"
f = glob.glob(out + "/stats/*/*kernel_stats.csv")[0]
print("== kernel stats")
for r in list(csv.DictReader(open(f)))[:10]:
    print("%-60s calls %6s avg %9.2f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(out + "/pmc_%s/*/*counter_collection.csv" % c)
    if not fs: continue
    # the force (non-energy) launches of the batched nonbonded kernel; the standalone ones of blues_batch_time_nonbonded come last
    rows = [r for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == c and "nonbonded" in r["Kernel_Name"] and "_b<" in r["Kernel_Name"] and "false" in r["Kernel_Name"]]
    by = {}
    for r in rows: by.setdefault(int(r["Dispatch_Id"]), 0.0); by[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    vals = [by[i] for i in sorted(by)[-50:]]
    res[c] = {"kernel": rows[0]["Kernel_Name"][:80] if rows else None, "launches": len(vals), "mean_KB": sum(vals) / max(1, len(vals))}
print("== pmc", json.dumps(res))
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
d = json.loads(open(out + "/bench.json").read())
print("== bench ns/day %.0f  ms/switch %.1f  K1 %.1f us frac %.4f single %s cpu %s" % (d["value"], d["ms_per_step"], d["roofline"]["usec_per_launch"], d["roofline"]["frac"], d.get("single_replica") and d["single_replica"]["value"], d.get("cpu_baseline") and d["cpu_baseline"]["value"]))
PY
