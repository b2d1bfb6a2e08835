#!/bin/bash
# usage (GPU box): bash scripts/evidence_r05_default.sh -- the default bench line with what has to agree with it, from ONE build, -> gpurun_out/ev_r05d/
#   bench_default.json               python bench.py (no flags: 2048 chains as two batches of 1024 taking turns; every chain its own start state; 3 timed switches)
#   bench_default_steps20.json       the command the round-end driver runs: --gpus 1 --steps 20 --warmup 5
#   bench_same_start.json            the same with --same-start (rounds 1-4: one common start state)
#   kernel_stats_default.csv         rocprofv3 --kernel-trace --stats of the same workload (1 warm-up + 1 timed switch per batch)
#   pmc_nonbonded.json               PMC counters of the nonbonded kernel, separate --pmc passes (scripts/pmc_nb.sh), keyed to the build's source hash
#   bench_default_with_counters.json the default line again once the counters are on disk (roofline.traffic / valu filled in)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ev_r05d; rm -rf $out; mkdir -p $out
python3 bench.py > $out/bench_default.log 2>&1; tail -1 $out/bench_default.log > $out/bench_default.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default_steps20.log 2>&1; tail -1 $out/bench_default_steps20.log > $out/bench_default_steps20.json
python3 bench.py --same-start --no-cpu --no-single > $out/bench_same_start.log 2>&1; tail -1 $out/bench_same_start.log > $out/bench_same_start.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-single > $out/stats.log 2>&1
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats_default.csv 2>/dev/null; rm -rf $out/stats
bash scripts/pmc_nb.sh r05 > $out/pmc.log 2>&1; cp gpurun_out/pmc_r05/pmc_nonbonded.json $out/pmc_nonbonded.json; cp $out/pmc_nonbonded.json profiles/r05_pmc_nonbonded.json
python3 bench.py --no-cpu --no-single > $out/bench_default_with_counters.log 2>&1; tail -1 $out/bench_default_with_counters.log > $out/bench_default_with_counters.json
python3 - <<PY
import json, csv
for f in ["bench_default", "bench_default_steps20", "bench_same_start", "bench_default_with_counters"]:
    try:
        d = json.loads(open("$out/%s.json" % f).read()); r = d["roofline"]
        print(f, round(d["value"]), d["steps"], round(d["ms_per_step"], 1), r["usec_per_launch"], r["frac"], r.get("traffic"), d["engine"]["setup_seconds"], d.get("single_replica") and d["single_replica"]["value"])
    except Exception as e:
        print(f, "failed", e)
for r in list(csv.DictReader(open("$out/kernel_stats_default.csv")))[:12]:
    print("   %-62s calls %6s avg %9.2f us  %5.1f%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
