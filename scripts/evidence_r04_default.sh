cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ev_r04b; rm -rf $out; mkdir -p $out
python3 bench.py > $out/bench_R1024.log 2>&1; tail -1 $out/bench_R1024.log > $out/bench_R1024.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-single > $out/stats.log 2>&1
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats_R1024.csv 2>/dev/null; rm -rf $out/stats
bash scripts/pmc_nb.sh r04 > $out/pmc.log 2>&1; cp gpurun_out/pmc_r04/pmc_nonbonded.json $out/pmc_nonbonded.json; cp $out/pmc_nonbonded.json profiles/r04_pmc_nonbonded.json
python3 bench.py --no-cpu --no-single > $out/bench_R1024_with_counters.log 2>&1; tail -1 $out/bench_R1024_with_counters.log > $out/bench_R1024_with_counters.json
python3 - <<PY
import json
for f in ["bench_R1024", "bench_R1024_with_counters"]:
    d = json.loads(open("$out/%s.json" % f).read()); r = d["roofline"]
    print(f, round(d["value"]), round(d["ms_per_step"], 1), r["usec_per_launch"], r["frac"], r.get("traffic"), d["engine"]["setup_seconds"], d.get("single_replica") and d["single_replica"]["value"])
PY
