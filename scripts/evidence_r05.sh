#!/bin/bash
# usage (GPU box): bash scripts/evidence_r05.sh [what ...]  -- round-5 evidence -> gpurun_out/ev_r05/ (copy what is to be judged into profiles/r05/)
#   full_R16 / full_R64     the FULL BLUES iteration (NCMC switch + Metropolis + reset + MD leg on the unfrozen System), md + alch + ncmc Simulations per chain
#   water_R1 / water_R16    configs[3] to the letter (2000-step switch, nothing frozen) through the fragment lists
#   stats_water_R16         rocprofv3 --kernel-trace --stats of the water_R16 workload
#   default                 `python bench.py` (chains decorrelated at set-up) and the same with --same-start
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ev_r05; mkdir -p $out
run() { name=$1; shift; timeout 1500 python3 bench.py "$@" > $out/$name.log 2>&1; tail -1 $out/$name.log > $out/$name.json; python3 - <<PY
import json
try:
    d = json.loads(open("$out/$name.json").read())
    r = d.get("roofline") or {}; f = d.get("full_iteration")
    print("== %-18s %9.0f ns/day  %8.1f ms/step  K1 %s us frac %s  setup %.1f s  boundary: %s" % ("$name", d["value"], d["ms_per_step"], r.get("usec_per_launch") and round(r["usec_per_launch"], 1), r.get("frac") and round(r["frac"], 4), d["engine"]["setup_seconds"], d["engine"]["plugin_boundary"]))
    if f: print("   full iteration: both legs %.0f ns/day, md leg %.0f | ms sync %.1f ncmc %.1f boundary %.1f md %.1f | us/chain-step md %.2f ncmc %.2f | %s" % (f["ns_day_both_legs"], f["ns_day_md_leg"], f["ms_sync"], f["ms_ncmc"], f["ms_boundary"], f["ms_md"], f["us_per_chain_step_md"], f["us_per_chain_step_ncmc"], f["md_engine"]))
    print("   engine:", {k: d["engine"][k] for k in ("seconds", "list_rebuilds_per_switch", "lockstep_steps_per_switch", "fallback_steps_per_switch")})
except Exception as e:
    print("== $name failed:", e); print(open("$out/$name.log").read()[-1500:])
PY
}
what="${@:-full_R16 full_R64 water_R1 water_R16 stats_water_R16 default}"
for w in $what; do case $w in
  full_R16) run full_R16 --md-steps 1000 --replicas 16 --groups 1 --steps 8 --warmup 2 --no-cpu ;;
  full_R64) run full_R64 --md-steps 1000 --replicas 64 --groups 1 --steps 8 --warmup 2 --no-cpu ;;
  full_R256) run full_R256 --md-steps 1000 --replicas 256 --groups 1 --steps 2 --warmup 2 --no-cpu --no-single ;;
  full_R1024) run full_R1024 --md-steps 1000 --replicas 1024 --groups 1 --steps 2 --warmup 2 --no-cpu --no-single ;;
  water_R1) run water_R1 --workload water --nsteps-nc 2000 --replicas 1 --groups 1 --steps 2 --warmup 1 --no-cpu --no-single ;;
  water_R16) run water_R16 --workload water --nsteps-nc 2000 --replicas 16 --groups 1 --steps 2 --warmup 1 --no-cpu --no-single ;;
  water_R64) run water_R64 --workload water --nsteps-nc 2000 --replicas 64 --groups 1 --steps 2 --warmup 1 --no-cpu --no-single ;;
  stats_water_R16)
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --workload water --nsteps-nc 2000 --replicas 16 --groups 1 --steps 1 --warmup 1 --no-cpu --no-single --no-kernel-timing > $out/stats_water.log 2>&1
    cp $out/stats/*/*kernel_stats.csv $out/kernel_stats_water_R16.csv 2>/dev/null; rm -rf $out/stats
    python3 - <<PY
import csv, os
p = "$out/kernel_stats_water_R16.csv"
if os.path.exists(p):
    for r in list(csv.DictReader(open(p)))[:14]:
        print("   %-62s calls %6s avg %9.2f us  %5.1f%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
    ;;
  default) run bench_default --no-cpu; run bench_same_start --same-start --no-cpu --no-single ;;
  sidechain_R64) run sidechain_R64 --workload sidechain --replicas 64 --groups 1 --nsteps-nc 5000 --steps 2 --warmup 1 --no-cpu --no-single ;;
esac; done
