cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
