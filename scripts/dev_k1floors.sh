#!/bin/bash
# usage (GPU box): bash scripts/dev_k1floors.sh [R] -- the nonbonded kernel alone at the initial geometry, for the default build and for
# every floor-experiment build blues_amd/csrc/libblues_hip_x*.so (-DK1X_<what>: kernels_nb.h; wrong numbers, only the time matters)
R=${1:-512}
cd $GRAFT_REPO_ROOT
python3 scripts/dev_k1exp.py $R 2>&1 | grep "pass"
for f in blues_amd/csrc/libblues_hip_x*.so; do BLUES_LIB_PATH=$PWD/$f python3 scripts/dev_k1exp.py $R 2>&1 | grep "pass"; done
