#!/bin/bash
# usage (GPU box): bash scripts/evidence_r04.sh -- the round's evidence in one call -> gpurun_out/ev_r04/ (copy what is to be judged into profiles/r04/)
#   bench_R2048_G2.json        the line of `python bench.py` (defaults: 2048 chains as two batches of 1024 taking turns on the device)
#   kernel_stats_R2048_G2.csv     rocprofv3 --kernel-trace --stats of the same workload (one timed switch)
#   pmc_nonbonded.json         PMC counters of the nonbonded kernel in separate --pmc passes (scripts/pmc_nb.sh), keyed to the build's source hash
#   water_R1 / water_R16 / sidechain_R64 / reciprocal_R512 / bench_R512 / bench_R2048 / bench_R2048_G4 .json   the other configurations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ev_r04; rm -rf $out; mkdir -p $out
run() { name=$1; shift; python3 bench.py "$@" > $out/$name.log 2>&1; tail -1 $out/$name.log > $out/$name.json; python3 - <<PY
import json
try:
    d = json.loads(open("$out/$name.json").read())
    r = d.get("roofline") or {}
    print("== %-16s %9.0f ns/day  %8.1f ms/switch  K1 %s us frac %s  setup %.1f s  single %s" % ("$name", d["value"], d["ms_per_step"], r.get("usec_per_launch") and round(r["usec_per_launch"], 1), r.get("frac") and round(r["frac"], 4), d["engine"]["setup_seconds"], d.get("single_replica") and round(d["single_replica"]["value"])))
except Exception as e:
    print("== $name failed:", e); print(open("$out/$name.log").read()[-1500:])
PY
}
if [ -z "$OTHERS_ONLY" ]; then   # (OTHERS_ONLY=1: only the other configurations below; the default line and what has to agree with it come from evidence_r04_default.sh)
run bench_R2048_G2
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-single > $out/stats.log 2>&1
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats_R2048_G2.csv 2>/dev/null; rm -rf $out/stats
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$out/kernel_stats_R2048_G2.csv")))[:14]:
    print("   %-62s calls %6s avg %9.2f us  %5.1f%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
bash scripts/pmc_nb.sh r04 > $out/pmc.log 2>&1; cp gpurun_out/pmc_r04/pmc_nonbonded.json $out/pmc_nonbonded.json 2>/dev/null; cp $out/pmc_nonbonded.json profiles/r04_pmc_nonbonded.json 2>/dev/null; tail -3 $out/pmc.log | head -2
run bench_R2048_G2_with_counters --no-cpu --no-single        # (the same line once the counters of this build are on disk: roofline.valu / traffic filled in)
fi
run water_R1 --workload water --replicas 1 --steps 2 --warmup 1 --no-cpu --no-single
run water_R16 --workload water --replicas 16 --groups 1 --steps 2 --warmup 1 --no-cpu --no-single
run sidechain_R64 --workload sidechain --replicas 64 --groups 1 --nsteps-nc 5000 --steps 2 --warmup 1 --no-cpu --no-single
run reciprocal_R512 --reciprocal --replicas 512 --groups 1 --steps 2 --warmup 1 --no-cpu --no-single
run bench_R512 --replicas 512 --groups 1 --no-cpu --no-single
run bench_R1024_steps3 --replicas 1024 --groups 1 --no-cpu --no-single
run bench_R2048 --replicas 2048 --groups 1 --steps 2 --no-cpu --no-single
run bench_R2048_G4 --replicas 2048 --groups 4 --concurrent --steps 2 --no-cpu --no-single     # four replica batches of 512 on four streams (the nonbonded kernel's duration then includes its co-runners)
