# usage (GPU box): bash scripts/r06_pytest.sh tag [pytest args] -- GPU tests, log under gpurun_out/r06/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p gpurun_out/r06
timeout 2400 python3 -m pytest tests -m gpu -q --durations=15 "$@" > gpurun_out/r06/pytest_$TAG.log 2>&1; echo "pytest rc $?"; tail -40 gpurun_out/r06/pytest_$TAG.log
