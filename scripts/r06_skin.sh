# usage (GPU box): bash scripts/r06_skin.sh -- bare stepping of 1024 chains against the outer list margin (BluesTuning.skin) and the inner one (prune_margin)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in "skin=0.20" "skin=0.24" "skin=0.28" "skin=0.32" "skin=0.24,prune_margin=0.05" "skin=0.28,prune_margin=0.06"; do
  echo "== $t"; BLUES_TUNING="assume_batch=1024,$t" timeout 600 python3 scripts/batch_scaling.py --nsteps 600 1024 2>&1 | grep 'us/step' | tail -1
done
