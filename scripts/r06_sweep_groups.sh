# usage (GPU box): bash scripts/r06_sweep_groups.sh -- bare stepping of 1024 chains under several tiles-per-list shapes (BluesTuning.list_group)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for t in "" "list_group=4" "list_group=3" "list_group=2" "list_group=1"; do
  echo "== [$t]"
  BLUES_TUNING=$t timeout 600 python3 scripts/batch_scaling.py --nsteps 600 1024 2>&1 | tail -2
done > gpurun_out/r06/sweep_groups.txt 2>&1
cat gpurun_out/r06/sweep_groups.txt
