# usage (GPU box): bash scripts/r06_forks.sh -- bare stepping of 1024 chains with the alchemical / bonded kernels on the main stream, beside the list builder, beside everything
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for t in "fork=1" "fork=3" "fork=2" "fork=0"; do
  echo "== [$t]"
  BLUES_TUNING=$t timeout 600 python3 scripts/batch_scaling.py --nsteps 600 1024 2>&1 | tail -1
done > gpurun_out/r06/forks.txt 2>&1
cat gpurun_out/r06/forks.txt
