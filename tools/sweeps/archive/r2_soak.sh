cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 || exit 1
timeout -k 10 400 python tests/fuzz_gpu.py 150 77 2>&1 | tail -4
