# usage (GPU box): bash scripts/dev_plainsweep.sh R nsteps "tuning1" "tuning2" ... -- step time of the batched stepping path under BLUES_TUNING settings (no profiler)
cd $GRAFT_REPO_ROOT
R=$1; NS=$2; shift 2
for t in "$@"; do
  BLUES_TUNING=$t python3 scripts/batch_scaling.py --nsteps $NS $R > /tmp/ps.log 2>&1; echo "[$t] $(grep 'us/step' /tmp/ps.log | tail -1)"
done
