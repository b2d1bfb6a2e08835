# usage (GPU box): bash scripts/r06_full.sh tag -- what the round-end driver does: the whole GPU suite, smoke(), then the bench command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-final}
mkdir -p gpurun_out/r06
timeout 2400 python3 -m pytest tests -m gpu -q --durations=10 > gpurun_out/r06/pytest_$TAG.log 2>&1; echo "pytest rc $?"; tail -16 gpurun_out/r06/pytest_$TAG.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke_$TAG.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r06/smoke_$TAG.log
bash scripts/r06_bench.sh $TAG
