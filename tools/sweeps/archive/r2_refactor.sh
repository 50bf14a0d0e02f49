cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 && VARIANTS="prev new" STEPS=30 EXTRA="--config 2" bash tools/sweeps/ab.sh
