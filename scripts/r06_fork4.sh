# usage (GPU box): bash scripts/r06_fork4.sh -- BluesTuning.fork = 4 against the default: the order test, then bare stepping of 1024 chains
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python3 -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "order_of_a_pass" > gpurun_out/r06/pytest_fork4.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06/pytest_fork4.log
for t in "assume_batch=1024,fork=4" "assume_batch=1024" "assume_batch=1024,fork=4" "assume_batch=1024,fork=3"; do
  echo "== $t"; BLUES_TUNING=$t timeout 600 python3 scripts/batch_scaling.py --nsteps 600 1024 2>&1 | grep 'us/step' | tail -1
done
bash scripts/r06_step.sh fork4 "assume_batch=1024,fork=4" | tail -14
