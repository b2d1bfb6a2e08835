#!/bin/bash
# usage (GPU box): bash scripts/evidence_r04_default.sh -- the default bench line with what has to agree with it, from ONE build, -> gpurun_out/ev_r04b/
#   bench_R2048_G2.json            python bench.py (no flags: 2048 chains as two batches of 1024 taking turns on the device; 3 timed switches)
#   bench_R2048_G2_steps20.json    the command the round-end driver runs: --gpus 1 --steps 20 --warmup 5
#   kernel_stats_R2048_G2.csv      rocprofv3 --kernel-trace --stats of the same workload (1 warm-up + 1 timed switch per batch)
#   pmc_nonbonded.json             PMC counters of the nonbonded kernel, separate --pmc passes (scripts/pmc_nb.sh), keyed to the build's source hash
#   bench_R2048_G2_with_counters.json   the default line again once the counters are on disk (roofline.traffic / valu filled in)
#   bench_R1024.json               one batch of 1024 chains (--groups 1), for comparison
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ev_r04b; rm -rf $out; mkdir -p $out
python3 bench.py > $out/bench_R2048_G2.log 2>&1; tail -1 $out/bench_R2048_G2.log > $out/bench_R2048_G2.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_R2048_G2_steps20.log 2>&1; tail -1 $out/bench_R2048_G2_steps20.log > $out/bench_R2048_G2_steps20.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-single > $out/stats.log 2>&1
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats_R2048_G2.csv 2>/dev/null; rm -rf $out/stats
bash scripts/pmc_nb.sh r04 > $out/pmc.log 2>&1; cp gpurun_out/pmc_r04/pmc_nonbonded.json $out/pmc_nonbonded.json; cp $out/pmc_nonbonded.json profiles/r04_pmc_nonbonded.json
python3 bench.py --no-cpu --no-single > $out/bench_R2048_G2_with_counters.log 2>&1; tail -1 $out/bench_R2048_G2_with_counters.log > $out/bench_R2048_G2_with_counters.json
python3 bench.py --replicas 1024 --groups 1 --steps 20 --warmup 5 --no-cpu --no-single > $out/bench_R1024.log 2>&1; tail -1 $out/bench_R1024.log > $out/bench_R1024.json
python3 - <<PY
import json, csv
for f in ["bench_R2048_G2", "bench_R2048_G2_steps20", "bench_R2048_G2_with_counters", "bench_R1024"]:
    d = json.loads(open("$out/%s.json" % f).read()); r = d["roofline"]
    print(f, round(d["value"]), d["steps"], round(d["ms_per_step"], 1), r["usec_per_launch"], r["frac"], r.get("traffic"), d["engine"]["setup_seconds"], d.get("single_replica") and d["single_replica"]["value"])
for r in list(csv.DictReader(open("$out/kernel_stats_R2048_G2.csv")))[:12]:
    print("   %-62s calls %6s avg %9.2f us  %5.1f%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
