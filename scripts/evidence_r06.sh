#!/bin/bash
# usage (GPU box): bash scripts/evidence_r06.sh [what ...] -- round-6 evidence from ONE build -> gpurun_out/ev_r06/ (copy what is to be judged into profiles/r06/)
#   default      python bench.py --gpus 1 --steps 20 --warmup 5 (the round-end driver's command)            -> bench_default_steps20.json
#   stats        rocprofv3 --kernel-trace --stats of the default workload (1 warm-up + 1 timed switch)      -> kernel_stats_default.csv
#   pmc          PMC counters of the nonbonded kernel, separate --pmc passes (scripts/pmc_nb.sh)              -> pmc_nonbonded.json (-> profiles/r06_pmc_nonbonded.json)
#   counters     the default command again once the counters are on disk                                      -> bench_default_with_counters.json
#   solute       --workload rotmove-solute (the mobile region of the reference's freeze_radius)               -> bench_rotmove_solute.json
#   full_R64 / full_R256   the FULL BLUES iteration (md + alch + ncmc per chain, MD leg), differential correction
#   water_R16    configs[3] to the letter
#   step         bare stepping of 1024 chains + its kernel table                                              -> step.txt, kernel_stats_step_R1024.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ev_r06; mkdir -p $out
run() { name=$1; shift; timeout 1500 python3 bench.py "$@" > $out/$name.log 2>&1; grep '^{' $out/$name.log | tail -1 > $out/$name.json; python3 - <<PY
import json
try:
    d = json.loads(open("$out/$name.json").read())
    r = d.get("roofline") or {}; f = d.get("full_iteration"); e = d["engine"]
    print("== %-28s %9.0f ns/day  %8.1f ms/step  K1 %s us frac %s counter-frac %s | max/median %s replans %s straggled %s resorts %s | setup %.1f s rss %.1f GiB failed %s" % ("$name", d["value"], d["ms_per_step"], r.get("usec_per_launch") and round(r["usec_per_launch"], 1), r.get("frac") and round(r["frac"], 4), r.get("frac_counter") and round(r["frac_counter"], 3),
          e.get("iteration_seconds_max_over_median") and round(e["iteration_seconds_max_over_median"], 3), e.get("replans"), e.get("straggled"), e.get("resorts"), e["setup_seconds"], d["memory"]["host_peak_rss_gib"], d["chains_failed"]))
    if f: print("   full iteration: both legs %.0f ns/day | ms sync %.1f ncmc %.1f boundary %.1f md %.1f | us/chain-step md %.2f ncmc %.2f | %s" % (f["ns_day_both_legs"], f["ms_sync"], f["ms_ncmc"], f["ms_boundary"], f["ms_md"], f["us_per_chain_step_md"], f["us_per_chain_step_ncmc"], f["md_engine"]))
except Exception as ex:
    print("== $name failed:", ex); print(open("$out/$name.log").read()[-1500:])
PY
}
what="${@:-default stats pmc counters solute}"
for w in $what; do case $w in
  default) run bench_default_steps20 --gpus 1 --steps 20 --warmup 5 ;;
  stats)
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-single > $out/stats.log 2>&1
    cp $out/stats/*/*kernel_stats.csv $out/kernel_stats_default.csv 2>/dev/null; rm -rf $out/stats
    python3 - <<PY
import csv, os
p = "$out/kernel_stats_default.csv"
if os.path.exists(p):
    for r in list(csv.DictReader(open(p)))[:12]:
        print("   %-62s calls %6s avg %9.2f us  %5.1f%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
    ;;
  pmc) bash scripts/pmc_nb.sh r06 > $out/pmc.log 2>&1; cp gpurun_out/pmc_r06/pmc_nonbonded.json $out/pmc_nonbonded.json && cp $out/pmc_nonbonded.json profiles/r06_pmc_nonbonded.json; tail -3 $out/pmc.log | cut -c1-300 ;;
  counters) run bench_default_with_counters --gpus 1 --steps 20 --warmup 5 --no-cpu ;;
  solute) run bench_rotmove_solute --workload rotmove-solute --steps 5 --warmup 2 --no-cpu --no-single ;;
  full_R64) run full_R64 --md-steps 1000 --replicas 64 --groups 1 --steps 8 --warmup 2 --no-cpu ;;
  full_R256) run full_R256 --md-steps 1000 --replicas 256 --groups 1 --steps 2 --warmup 2 --no-cpu --no-single ;;
  full_R1024) run full_R1024 --md-steps 1000 --replicas 1024 --groups 1 --steps 2 --warmup 2 --no-cpu --no-single ;;
  water_R16) run water_R16 --workload water --nsteps-nc 2000 --replicas 16 --groups 1 --steps 2 --warmup 1 --no-cpu --no-single ;;
  sidechain_R64) run sidechain_R64 --workload sidechain --replicas 64 --groups 1 --nsteps-nc 5000 --steps 2 --warmup 1 --no-cpu --no-single ;;
  step) bash scripts/r06_step.sh final "assume_batch=1024" > $out/step.txt 2>&1; cp gpurun_out/r06/step_final/kernel_stats.csv $out/kernel_stats_step_R1024.csv 2>/dev/null; cat $out/step.txt ;;
esac; done
