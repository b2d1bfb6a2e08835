# usage (GPU box): bash scripts/r06_pytest_k.sh tag "<-k expression>" -- selected GPU tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1
mkdir -p gpurun_out/r06
timeout 1500 python3 -m pytest tests -m gpu -q --durations=8 -k "$2" > gpurun_out/r06/pytest_$TAG.log 2>&1; echo "pytest rc $?"; tail -40 gpurun_out/r06/pytest_$TAG.log
