# usage (GPU box): bash scripts/r06_groups.sh -- the bench line with 2, 3 and 4 batches of 1024 chains taking turns on the device
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for cfg in "3072 3" "4096 4"; do
  set -- $cfg
  timeout 900 python3 bench.py --replicas $1 --groups $2 --steps 8 --warmup 2 --no-cpu --no-single > gpurun_out/r06/bench_G$2.json 2> gpurun_out/r06/bench_G$2.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r06/bench_G$2.json')); e=d['engine']; print('G=$2 R=$1: %.1f k ns/day, %.1f ms/iteration, K1 %.1f us, setup %.1f s, hbm %.0f GiB, rss %.1f GiB, max/median %.3f' % (d['value']/1e3, d['ms_per_step'], d['roofline']['usec_per_launch'], e['setup_seconds'], d['memory']['device_in_use_gib'], d['memory']['host_peak_rss_gib'], e['iteration_seconds_max_over_median']))"
done
