# usage (GPU box): bash scripts/r06_pack.sh -- the packed step kernel and the one-path constraint solver: parity tests, then bare stepping of 1024 chains with and without the packing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1200 python3 -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -x -q -m gpu -k "full_size_batch_of_eight or fused_and_separate or order_of_a_pass or parity or other_programs or constraint or teacher" > gpurun_out/r06/pytest_pack.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r06/pytest_pack.log
bash scripts/r06_step.sh pack "assume_batch=1024"
bash scripts/r06_step.sh nopack "assume_batch=1024,pack_clusters=1" | head -8
